"""What bounds gemm_f32_split_bf16?  (GPU box; diagnostic builds with WRONG results, timing only.)  Variants of the library:
  frag   -DT2_GEMM_ABL_FRAG   half of the LDS fragment reads (the second 32-row tile of each operand reuses the first one's registers)
  store  -DT2_GEMM_ABL_STORE  one bf16 plane stored to LDS instead of three (the split arithmetic itself is kept)
  both
Shapes: the forward's pre_dec (NT), the backward's dxdec (NN) and dW_hh (TN split-K)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tacotron2_amd.build import build, build_variant
CODE = """
import sys, os; sys.path.insert(0, %r)
import tools.bench_gemm_cases as c
""" % ROOT
variants = [("product", None), ("frag", ["T2_GEMM_ABL_FRAG"]), ("store", ["T2_GEMM_ABL_STORE"]), ("both", ["T2_GEMM_ABL_FRAG", "T2_GEMM_ABL_STORE"])]
for name, defs in variants:
    lib = build(verbose=False) if defs is None else build_variant("gemm_" + name, defs, verbose=False)
    env = dict(os.environ, T2_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_gemm_cases.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(f"--- {name} ---")
    print(r.stdout.strip() if r.returncode == 0 else r.stderr[-1500:], flush=True)
