import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.engine import gemm
dev = torch.device("cuda:0")
M, N, K = 128, 128, 32
A = torch.zeros(M, K)
for k in range(K):
    A[k, k] = 1.0
B = torch.zeros(K, N)
for k in range(K):
    for n in range(N):
        B[k, n] = 256 * k + n      # exact in bf16x3 (< 2^24)
C = torch.full((M, N), float("nan"), device=dev)
gemm(A.to(dev), B.to(dev), C, M, N, K, K, N, N, a_k=1, b_k=0)
torch.cuda.synchronize()
C = C.cpu()
torch.set_printoptions(linewidth=250, sci_mode=False)
print("C[k, n] should be 256k + n; printing (C // 256, C % 256) for k < 18, n in 0..5 and 16..18, 32, 33, 64, 127")
cols = [0, 1, 2, 3, 4, 5, 16, 17, 18, 32, 33, 64, 127]
for k in range(18):
    print(k, [(int(C[k, n]) // 256, int(C[k, n]) % 256) for n in cols])
