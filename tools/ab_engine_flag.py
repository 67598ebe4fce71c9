"""A/B of Engine attributes on the bench batch (GPU box).  Usage:
    python tools/ab_engine_flag.py "attr=v0" "attr=v1,attr2=w" ...        (each argument = one variant; "base" = defaults)
    python tools/ab_engine_flag.py --hiprio "base" ...                    (the training step on a high-priority stream)
Variants alternate in one process (two repetitions, 3 warm-up + 8 timed steps each); prints ms per step and the HIP-event segments."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
from tacotron2_amd.trainer import Trainer
args = sys.argv[1:]
hiprio = "--hiprio" in args
args = [a for a in args if a != "--hiprio"]
dev = torch.device("cuda:0")
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
defaults = {}
import contextlib
stream_ctx = (lambda: torch.cuda.stream(hp)) if hiprio else contextlib.nullcontext
if hiprio:
    hp = torch.cuda.Stream(device=dev, priority=-1)
for rep in range(2):
    for var in args:
        for k, v in defaults.items():
            setattr(tr.engine, k, v)
        if var != "base":
            for kv in var.split(","):
                k, v = kv.split("=")
                assert hasattr(tr.engine, k), k
                defaults.setdefault(k, getattr(tr.engine, k))
                setattr(tr.engine, k, eval(v))
        with stream_ctx():
            for _ in range(3):
                tr.train_step(batch)
            torch.cuda.synchronize()
            tr.engine.profile = True
            t0 = time.perf_counter()
            for _ in range(8):
                loss3, _ = tr.train_step(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 8 * 1e3
            tr.engine.profile = False
        seg = tr.engine.segment_times_ms()
        keys = ("fwd.enc.convs", "fwd.enc.bilstm", "fwd.dec.attn_chain", "fwd.postnet", "bwd.postnet", "bwd.dec.chains", "bwd.bilstm", "bwd.encoder_convs")
        print(f"{var}{' [hiprio]' if hiprio else ''}: {dt:.2f} ms/step  " + "  ".join(f"{k} {seg.get(k, 0):.2f}" for k in keys) + f"  loss {float(loss3.sum()):.4f}", flush=True)
