"""Three shapes of the training step through t2_gemm (run by tools/ablate_gemm.py under T2_LIB_PATH)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.engine import gemm, splitk_for
dev = torch.device("cuda:0")


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, M, N, K, a_k, b_k, splitk=1):
    A = torch.randn((M, K) if a_k else (K, M), device=dev)
    B = torch.randn((N, K) if b_k else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    ms = timed(lambda: gemm(A, B, C, M, N, K, K if a_k else M, K if b_k else N, N, a_k=a_k, b_k=b_k, accumulate=2 if splitk > 1 else 0, splitk=splitk))
    print(f"{name:28s} {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)


R = 27904
case("pre_dec full (NT)", R, 4096, 1536, 1, 1)
case("postnet conv (NT)", 28032, 512, 2560, 1, 1)
case("dxdec full (NN)", R, 1536, 4096, 1, 0)
case("dW_hh_dec (TN split-K)", 4096, 1024, R, 0, 0, splitk_for(4096, 1024, R))
