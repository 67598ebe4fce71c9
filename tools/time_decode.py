"""Frame-loop time of the autoregressive decode (GPU box): python tools/time_decode.py [frames] [batch sizes...]."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import VANILLA
from tacotron2_amd.engine import Engine
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ps = ParamStore(VANILLA, dev); init_parameters(ps, 0)
eng = Engine(ps)
for B in ([int(x) for x in sys.argv[2:]] or [64, 32]):
    ib = ljspeech_batch(B, seed=4321, num_speakers=4)
    ci, cl, spk = ib["chars_idx"].to(dev), ib["chars_idx_len"].to(dev), ib["speaker_id"].to(dev)
    eng.infer(ci, cl, 32, speaker_id=spk, training=False, seed=1)
    torch.cuda.synchronize()
    for rep in range(2):
        eng.profile = True; eng.marks = []; eng.spans = []
        eng.mark("inf.start")
        t0 = time.perf_counter()
        out = eng.infer(ci, cl, n, speaker_id=spk, training=False, seed=2, check_every=64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.profile = False
        seg = eng.segment_times_ms()
        frames = int(out[0].shape[1])
        print(f"B={B} frames={frames} call {dt * 1e3:.1f} ms  frame_loop {seg.get('inf.frame_loop', 0):.2f} ms = "
              f"{seg.get('inf.frame_loop', 0) * 1e3 / max(frames, 1):.2f} us/step  encoder {seg.get('inf.encoder', 0):.2f} ms", flush=True)
