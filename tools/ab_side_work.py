"""Timing-only ablation (GPU box): the training step with the side-stream GEMMs skipped, to see how much the latency-bound
chains on the main stream are stretched by the GEMM workgroups that share the CUs.  (Gradients are wrong in the ablated runs.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
import tacotron2_amd.engine as E
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
from tacotron2_amd.trainer import Trainer
dev = torch.device("cuda:0")
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
SKIP = [False]
real_gemm = E.gemm
def gemm(*a, **k):
    if SKIP[0] and tr.engine._side is not None and torch.cuda.current_stream() == tr.engine._side:
        return
    return real_gemm(*a, **k)
E.gemm = gemm
for rep in range(2):
    for skip in (False, True):
        SKIP[0] = skip
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        tr.engine.profile = True
        t0 = time.perf_counter()
        for _ in range(8):
            tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8 * 1e3
        tr.engine.profile = False
        seg = tr.engine.segment_times_ms()
        print(f"skip_side_gemms={skip}: {dt:.2f} ms/step  " + "  ".join(f"{k} {v:.2f}" for k, v in seg.items() if v > 0.3), flush=True)
