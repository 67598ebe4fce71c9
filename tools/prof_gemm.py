"""One GEMM shape, a few launches (for rocprofv3 --pmc passes): python tools/prof_gemm.py M N K a_k b_k [splitk] [iters]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.engine import gemm
M, N, K, a_k, b_k = (int(x) for x in sys.argv[1:6])
splitk = int(sys.argv[6]) if len(sys.argv) > 6 else 1
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device("cuda:0")
A = torch.randn((M, K) if a_k else (K, M), device=dev)
B = torch.randn((N, K) if b_k else (K, N), device=dev)
C = torch.zeros(M, N, device=dev)
for _ in range(iters):
    gemm(A, B, C, M, N, K, K if a_k else M, K if b_k else N, N, a_k=a_k, b_k=b_k, accumulate=2 if splitk > 1 else 0, splitk=splitk)
torch.cuda.synchronize()
