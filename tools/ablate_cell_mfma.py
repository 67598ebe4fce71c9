"""What could a faster matrix path buy the LSTM step kernels?  (GPU box; diagnostic, timing only - the ablated builds compute wrong
results.)  Builds libtacotron2_amd_keep<N>.so with -DT2_CELL_MFMA_KEEP=N (N of the 4 fp32 MFMA k-substeps of every chunk issued; same
loads, same epilogue, same launches) and times, per build in its own process: the autoregressive frame loop at 64 and 32 utterances
(tools/time_decode.py) and the training step at 32 and 64 utterances per GPU (bench.py).  Six exact bf16 products on the bf16 pipe
take 0.375 of the fp32 MFMA time, i.e. between N = 2 and N = 1 - WITHOUT the operand-split VALU work and the 1.5x weight bytes such a
path adds, so these numbers are an upper bound of its gain."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tacotron2_amd.build import build_variant

for keep in (4, 2, 1):
    lib = build_variant(f"keep{keep}", [f"T2_CELL_MFMA_KEEP={keep}"], verbose=False)
    env = dict(os.environ, T2_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_decode.py"), "400", "64", "32"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    dec = [l for l in r.stdout.splitlines() if l.startswith("B=")]
    print(f"keep {keep}/4 MFMA substeps:", flush=True)
    for l in dec[1::2]:
        print("   decode  " + l, flush=True)
    if r.returncode != 0:
        print(r.stderr[-1500:])
    for b in (32, 64):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--batch", str(b),
                            "--no-cpu-baseline", "--no-decode", "--no-high"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if js:
            d = json.loads(js[-1]); s = d["segments_ms"]
            print(f"   train b={b}: {d['ms_per_step']:.2f} ms/step  fwd attention chain {s.get('fwd.dec.attn_chain', 0):.2f}  "
                  f"bwd chains {s.get('bwd.dec.chains', 0):.2f}", flush=True)
        else:
            print(r.stderr[-1500:])
