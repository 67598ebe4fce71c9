"""Phase stamps of the backward attention kernels inside a real training step (diagnostic, GPU box only)."""
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
batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
clk = torch.zeros(32, dtype=torch.int64, device=dev)
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
