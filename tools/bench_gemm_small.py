"""Split-K on the under-filled forward / data-gradient GEMM shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("bg", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_gemm.py"))
src = open(spec.origin).read().split("\nR = 27904\n")[0]      # helpers only, not the shape list
ns = {"__name__": "bg", "__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
case = ns["case"]
for sk in (1, 2, 3, 4):
    case("encoder conv fwd (NT)", 6016 + 124, 512, 2560, 1, 1, sk)
for sk in (1, 2, 3, 4):
    case("encoder conv dgrad (NT)", 6016 + 124, 512, 2560, 1, 1, sk)
for sk in (1, 2, 4):
    case("proj dgrad (NN) M=27904 N=1536 K=81->96", 27904, 1536, 96, 1, 0, sk)
for sk in (1, 2):
    case("bilstm dx (NN)", 6016, 512, 2048, 1, 0, sk)
for sk in (1, 2, 4):
    case("dmem ctx (TN batched-ish one sample)", 188, 512, 872, 0, 0, sk)
