"""A/B of the backward schedule on the bench batch (GPU box): two-stream pipeline (decoder-LSTM BPTT as its own launches on the
side stream) against the hosted schedule (Engine.bwd_host: the BPTT steps ride in the attention chain's products launches),
for several BPTT chunk sizes / margins.  One process, alternating variants, two repetitions."""
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
variants = [(False, 80, 0), (True, 48, 8), (True, 32, 8), (True, 64, 10), (True, 80, 12), (True, 48, 16)]
if len(sys.argv) > 1:
    variants = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    variants = [(bool(h), c, m) for h, c, m in variants]
for rep in range(2):
    for host, chunk, margin in variants:
        e = tr.engine
        e.bwd_host = host
        if host:
            e.bwd_host_chunk, e.bwd_host_margin = chunk, margin
        else:
            e.chunk_bwd = chunk
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        e.profile = True
        t0 = time.perf_counter()
        for _ in range(8):
            loss3, _ = tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8 * 1e3
        e.profile = False
        seg = e.segment_times_ms()
        print(f"bwd_host={host} chunk={chunk} margin={margin}: {dt:.2f} ms/step  bwd.dec.chains {seg.get('bwd.dec.chains', 0):.2f}  "
              f"attn_gemms {seg.get('bwd.dec.attn_gemms', 0):.2f}  bilstm {seg.get('bwd.bilstm', 0):.2f}  "
              f"loss {float(loss3.sum()):.4f}", flush=True)
