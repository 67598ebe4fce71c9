"""A/B of the forward decoder-LSTM chain on the bench batch (GPU box): persistent weight-stationary launches on the side stream
vs one launch per frame there (the round-1 variant with the steps hosted in the attention-energies launches was removed in
round 4; its record: profiles/r02_ab_fwd_dec_chain.txt)."""
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
    for mode, chunk, gs in (("steps", 64, True), ("persistent", 64, False), ("persistent", 64, True), ("persistent", 96, True)):
        tr.engine.dec_chain, tr.engine.chunk, tr.engine.persist_gemm_side = mode, chunk, gs
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        tr.engine.check_persistent_kernels()
        tr.engine.profile = True
        t0 = time.perf_counter()
        for _ in range(8):
            loss3, _ = tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8 * 1e3
        tr.engine.profile = False
        seg = tr.engine.segment_times_ms()
        fwd = sum(v for k, v in seg.items() if k.startswith("fwd.dec."))
        print(f"dec_chain={mode:10s} chunk={chunk:3d} gemm_side={gs}: {dt:.2f} ms/step  fwd.dec.* {fwd:.2f}  attn_chain {seg.get('fwd.dec.attn_chain', 0):.2f}  "
              f"tail {seg.get('fwd.dec.lstm_chain_tail', 0):.2f}  loss {float(loss3.sum()):.5f}", flush=True)
