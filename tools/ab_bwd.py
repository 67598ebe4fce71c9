"""A/B of the backward schedule on the bench batch (GPU box): chunk size of the two-stream backward pipeline (round 2 also ran
this script with the decoder BPTT hosted in the attention cell-backward launches: profiles/r02_ab_bwd_schedule.txt)."""
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
for rep in range(2):
    for co, chunk in ((False, 80), (False, 48), (False, 64), (False, 112), (False, 160), (False, 218)):
        tr.engine.chunk_bwd = chunk
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
        print(f"chunk_bwd={chunk}: {dt:.2f} ms/step  bwd.dec.chains {seg.get('bwd.dec.chains', 0):.2f}  "
              f"attn_gemms {seg.get('bwd.dec.attn_gemms', 0):.2f}  bilstm {seg.get('bwd.bilstm', 0):.2f}", flush=True)
