import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tacotron2_ref as R
from tacotron2_amd import engine
from tests.test_gpu_model import build_engine, masks_to_device, random_case
dev = torch.device("cuda:0")
d = R.default_dims(num_chars=39, encoded_dim=64, prenet_dim=32, att_rnn_dim=64, att_dim=32, rnn_hidden_dim=64, postnet_dim=64, num_mels=16, dropout=0.5)
P = R.init_params(d, seed=11)
B, L = 35, 17
ci, lens, mel, tl, gate, masks = random_case(d, B, L, 23, 77, dev)
snaps = {}
for native in (1, 0, 2, 3):
    engine.GEMM_NATIVE_FP32[0] = native & 1
    eng, ps = build_engine(d, P, dev)
    eng.chunk, eng.chunk_bwd = 7, 9
    outs, ctx = eng.forward_tf(ci.to(dev), lens.to(dev), mel.to(dev), tl.to(dev), training=True, masks=masks_to_device(masks, dev))
    ps.grad.zero_()
    eng.loss_and_grads(outs, ctx, mel.to(dev), gate.to(dev))
    torch.cuda.synchronize()
    snaps[native] = {k: v.clone() for k, v in eng._ws.items() if v.dtype == torch.float32}
def indiff(x, y, name):
    a, b = snaps[x][name].double(), snaps[y][name].double()
    rows, C = 731, 64
    a, b = a[:rows * C].view(rows, C), b[:rows * C].view(rows, C)
    return float((a - b).abs().max()), float(a.abs().max())
for name in ("enc.dx3", "enc.conv2.draw", "enc.conv2.dx", "enc.conv1.raw", "enc.conv1.draw", "enc.conv1.dx", "enc.conv0.draw"):
    print(f"{name:18s} in-range max diff: native vs split {indiff(1, 0, name)[0]:.3e}   split vs split {indiff(0, 2, name)[0]:.3e}   native vs native {indiff(1, 3, name)[0]:.3e}   scale {indiff(1, 0, name)[1]:.3e}")
