import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine
from tacotron2_amd.engine import gemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (M, N, K, lda) in [(731, 64, 320, 64), (731, 64, 320, 320), (640, 64, 320, 64), (731, 128, 320, 64), (91, 64, 320, 64), (200, 64, 320, 64)]:
    rows = M if lda >= K else M + 4
    A = torch.randn(rows * (lda if lda < K else K) if lda < K else M * K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    if lda < K:
        Amat = torch.stack([A[r * lda: r * lda + K] for r in range(M)])
    else:
        Amat = A.view(M, K)
    ref = Amat.double() @ Bm.double().t()
    for native in (0, 1):
        engine.GEMM_NATIVE_FP32[0] = native
        C = torch.full((M, N), float("nan"), device=dev)
        gemm(A.to(dev), Bm.to(dev), C, M, N, K, lda, K, N)
        torch.cuda.synchronize()
        err = (C.double().cpu() - ref).abs()
        bad_rows = (err.max(1).values > 1e-3).nonzero().flatten().tolist()
        print(f"M={M} N={N} K={K} lda={lda} native={native}: max err {float(err.max()):.3e}; bad rows {bad_rows[:10]} ... n={len(bad_rows)}")
