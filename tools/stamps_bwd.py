"""Phase stamps of the backward frame chain inside a real training step (diagnostic build, GPU box only): the four dependent
launches of a frame - products of dgates[t+1] (192 workgroups), attention dw, attention ds, cell backward - on ONE time axis
(100 MHz wall clock at entry / exit of workgroup (0,0,0)), with the phase stamps of each."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.build import build_stamps
os.environ["T2_LIB_PATH"] = build_stamps()      # the diagnostic library: phase stamps are compiled out of the product build
import bench
from tacotron2_amd.params import ParamStore
from tacotron2_amd.trainer import Trainer
from tacotron2_amd.synthetic import ljspeech_batch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from tacotron2_amd.init import init_parameters
ps = ParamStore(bench.VANILLA, dev)
init_parameters(ps, seed=0)
tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, scheduler_milestones=(50000, 75000))
batch = {k: v.to(dev) for k, v in ljspeech_batch(int(os.environ.get("T2_STAMP_BATCH", "32")), seed=1234, num_speakers=4).items()}
clk = torch.zeros(128, dtype=torch.int64, device=dev)
for i in range(3):
    if i == 2:
        tr.engine.clk_bwd = clk
    tr.train_step(batch)
torch.cuda.synchronize()
c = clk.cpu().tolist()
GHZ = 2.38
f = lambda a, b0: (c[a] - c[b0]) / GHZ / 1e3
print("dw kernel, workgroup (0,0), us from entry: small loads consumed %.2f | sigma %.2f | exit %.2f" % (f(17, 16), f(18, 16), f(19, 16)))
# (matrix-pipe build, L <= 252: stamp 28 = dU MFMAs done, 27 = d_in MFMAs done and every wave past the barrier, 29 = K shares summed)
print("ds kernel, workgroup (0,0), us from entry: staged %.2f | phases A/B (ds, dpmT, dq, dv) %.2f | dU products %.2f | d_in products "
      "+ barrier %.2f | shares reduced %.2f | exit %.2f" % (f(25, 24), f(26, 24), f(28, 24), f(27, 24), f(29, 24), f(30, 24)))
# ---- event ring: the last 11 stamped launches of the chain (the end of the backward frame loop: frames 2, 1, 0) ----
KIND = {0: "products (dctx_tot, dh_rec; 192 wgs, K = 4096)", 1: "cell backward (dq.Wq + pointwise; 128 wgs, K = 128)",
        2: "BPTT step", 3: "attention dw", 4: "attention ds (matrix pipe)"}
ev = []
for s in range(11):
    e = c[40 + 8 * s: 48 + 8 * s]
    if e[1]:
        ev.append(e)
ev.sort(key=lambda e: e[1])
print(f"event ring: {c[32]} stamped launches in the step, the last {len(ev)} (wall clock 100 MHz -> 0.01 us; shader clock {GHZ} GHz):")
prev_exit = None
for e in ev:
    dur = (e[2] - e[1]) / 100.0
    gap = "" if prev_exit is None else f"  gap from previous exit {(e[1] - prev_exit) / 100.0:5.2f} us"
    sh = lambda i: (e[i] - e[3]) / GHZ / 1e3
    ph = ""
    if e[0] in (0, 1, 2):
        ph = f"  first operands consumed {sh(4):.2f} | main loop end {sh(5):.2f} | K shares in LDS {sh(6):.2f} | epilogue stored {sh(7):.2f}"
        if e[4] == 0:      # (short K: a single group, no stamp inside the loop)
            ph = f"  main loop end {sh(5):.2f} | K shares in LDS {sh(6):.2f} | epilogue stored {sh(7):.2f}"
    print(f"  {KIND.get(e[0], e[0]):52s} in-kernel {dur:5.2f} us{gap}{ph}")
    prev_exit = e[2]
