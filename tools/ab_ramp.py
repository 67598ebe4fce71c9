"""A/B of one boolean engine attribute on the bench batch (GPU box): python tools/ab_ramp.py [attribute, default ramp_chunks]."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
from tacotron2_amd.trainer import Trainer
dev = torch.device("cuda:0")
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
ATTR = sys.argv[1] if len(sys.argv) > 1 else "ramp_chunks"
for rep in range(3):
    for ramp in (True, False):
        setattr(tr.engine, ATTR, ramp)
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        tr.engine.profile = True
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        tr.engine.profile = False
        seg = tr.engine.segment_times_ms()
        tr.engine.check_persistent_kernels()
        print(f"{ATTR}={ramp}: {dt:.2f} ms/step  bwd.postnet {seg['bwd.postnet']:.2f}  bilstm {seg['bwd.bilstm']:.2f}  fwd chain {seg['fwd.dec.attn_chain']:.2f} tail {seg['fwd.dec.lstm_chain_tail']:.2f}  "
              f"bwd chains {seg['bwd.dec.chains']:.2f}", flush=True)
