"""A/B (GPU box): number of pipeline chunks per weight-gradient GEMM call in the backward frame loop (engine.wgrad_group)."""
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
    for wg, cb in ((4, 80), (5, 64), (6, 48), (3, 96), (3, 112), (2, 160)) if os.environ.get("T2_SWEEP_CHUNK") else [(w, 80) for w in (0, 1, 2, 3, 4, 6)]:
        tr.engine.chunk_att_wgrads = wg > 0
        tr.engine.wgrad_group = max(wg, 1)
        tr.engine.chunk_bwd = cb
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
        print(f"chunk_bwd={cb} wgrad_group={wg} (0 = all at the end): {dt:.2f} ms/step  chains {seg['bwd.dec.chains']:.2f}  bilstm {seg['bwd.bilstm']:.2f}  "
              f"convs {seg['bwd.encoder_convs']:.2f}", flush=True)
