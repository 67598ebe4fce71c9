"""Standalone throughput of t2_gemm on the shapes of one training step (GPU box only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.engine import gemm, splitk_for, set_float32_matmul_precision
set_float32_matmul_precision(os.environ.get("T2_MATMUL_PRECISION", "highest"))     # "high": three of the six bf16 products

dev = torch.device("cuda:0")


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, M, N, K, a_k, b_k, splitk=1):
    # A is [M][K] when a_k else [K][M]; B is [N][K] when b_k else [K][N]
    mk = torch.zeros if os.environ.get("T2_BENCH_ZEROS") else torch.randn     # zeros: the clock the chip holds without data toggling
    A = mk((M, K) if a_k else (K, M), device=dev)
    B = mk((N, K) if b_k else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    lda = K if a_k else M
    ldb = K if b_k else N
    acc = 2 if splitk > 1 else 0
    ms = timed(lambda: gemm(A, B, C, M, N, K, lda, ldb, N, a_k=a_k, b_k=b_k, accumulate=acc, splitk=splitk))
    print(f"{name:44s} M={M:6d} N={N:5d} K={K:6d} a_k={a_k} b_k={b_k} splitk={splitk:2d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)


R = 27904
case("pre_dec full (NT)", R, 4096, 1536, 1, 1)
case("pre_dec chunk 64 (NT)", 2048, 4096, 1536, 1, 1)
case("pre_att (NT)", R, 4096, 256, 1, 1)
case("postnet conv 512->512 (NT)", 28032, 512, 2560, 1, 1)
case("dxdec chunk 64 (NN)", 2048, 1536, 4096, 1, 0)
case("dxdec chunk 128 (NN)", 4096, 1536, 4096, 1, 0)
case("dxdec full (NN)", R, 1536, 4096, 1, 0)
case("dW_ih_dec (TN split-K)", 4096, 1536, R, 0, 0, splitk_for(4096, 1536, R))
case("dW_hh_dec (TN split-K)", 4096, 1024, R, 0, 0, splitk_for(4096, 1024, R))
case("dW_hh_dec (TN split-K 4)", 4096, 1024, R, 0, 0, 4)
case("dW_ih_att ctx (TN split-K)", 4096, 512, R, 0, 0, splitk_for(4096, 512, R))
case("dW postnet (TN split-K)", 512, 2560, 28032, 0, 0, splitk_for(512, 2560, 28032))
case("postnet dgrad (NT)", 28032, 512, 2560, 1, 1)

print("--- split-K sweeps ---")
for sk in (2, 3, 4, 6, 7, 8, 12):
    case("dW postnet (TN)", 512, 2560, 28032, 0, 0, sk)
for sk in (1, 2, 3, 4):
    case("dW_hh_dec (TN)", 4096, 1024, R, 0, 0, sk)
for sk in (1, 2, 4):
    case("dxdec chunk 80 (NN)", 2560, 1536, 4096, 1, 0, sk)
for sk in (2, 4, 8):
    case("dW_q (TN)", 128, 1024, R, 0, 0, sk)
for sk in (4, 8, 16, 28):
    case("dW prenet2 (TN)", 256, 256, R, 0, 0, sk)
case("encoder conv (NT)", 6016 + 128, 512, 2560, 1, 1)
case("bilstm in-proj (NT)", 6016, 2048, 512, 1, 1)
case("att_encoder batched-ish (NT)", 6016, 128, 512, 1, 1)
