"""Attention chain (forward) timing with and without the backward stashes (GPU box only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.build import build_stamps
os.environ["T2_LIB_PATH"] = build_stamps()      # the diagnostic library: phase stamps are compiled out of the product build
from tacotron2_amd import _lib
from tacotron2_amd._lib import call, make

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, L, T, A, Ad, Ef, KL = 32, 188, 128, 1024, 128, 512, 31


def pack(W, H):
    K = W.shape[1]
    arr = (_lib.S["T2Seg"] * 1)()
    arr[0].w = W.data_ptr(); arr[0].ldw = K; arr[0].K = K
    ntpad = (K // 16 + 15) // 16 * 16
    out = torch.empty(H // 4 * ntpad * 256, device=dev)
    call("t2_lstm_pack_fwd", arr, 1, H, out, st)
    return out


W = torch.randn(4 * A, A + Ef, device=dev) / 40
wp = pack(W, A)
Wq = torch.randn(Ad, A, device=dev) / 32
U = torch.randn(Ad, 2, KL, device=dev) / 8
v = torch.randn(Ad, device=dev) / 11
pre = torch.randn(T, B, 4 * A, device=dev)
pmT = torch.randn(B, Ad, L, device=dev)
memory = torch.randn(B, L, Ef, device=dev)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
xdec = torch.zeros(T + 1, B, A + Ef, device=dev)
Bp = 32
xdec_t = torch.zeros(T + 1, (A + Ef) // 16, Bp, 16, device=dev)
att_c = torch.zeros(T + 1, B, A, device=dev)
gates = torch.empty(T, B, 4 * A, device=dev)
align = torch.empty(B, T, L, device=dev)
cum = torch.zeros(T + 1, B, L, device=dev)
th = torch.empty(T, B, Ad, (L + 3) // 4 * 4, device=dev)
xproj = torch.zeros(T + 1, B, 1024 + Ef, device=dev)
e_part = torch.empty(B, Ad // 16, L, device=dev)


clk = torch.zeros(32, dtype=torch.int64, device=dev)


def run(use_th, use_gates, stamps=False):
    seq = make("T2AttnSeq", B=B, L=L, T=T, A=A, Ad=Ad, Ef=Ef, Kl=KL, wpacked=wp, W_ih_ctx=W, ld_wih=A + Ef, W_hh=W, Wq=Wq, U=U, v=v,
               pre=pre, pmT=pmT, memory=memory, len=lens, xdec=xdec, att_c=att_c, gates=gates if use_gates else None,
               align=align, cum=cum, th=th if use_th else None, xproj_ctx=xproj[1:, :, 1024:].data_ptr(), ld_xproj=1024 + Ef,
               e_part=e_part, xdec_t=xdec_t, clk=clk if stamps else None)
    call("t2_attn_seq_fwd", seq, st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); call("t2_attn_seq_fwd", seq, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / T


for use_th, use_gates in [(True, True), (False, True), (True, False), (False, False)]:
    print(f"th stash {use_th}, gates stash {use_gates}: {run(use_th, use_gates):.2f} us per frame (cell + energies + context)", flush=True)

import ctypes
out8 = (ctypes.c_uint64 * 8)()
_lib.lib().t2_debug_clock(1, out8)
run(True, True, stamps=True)
_lib.lib().t2_debug_clock(0, out8)
cc = list(out8)
c = clk.cpu().tolist()
GHZ = 2.38   # shader clock during the loop (s_memtime ticks / s_memrealtime), DESIGN.md section 4.1
e = [(c[i] - c[0]) / GHZ / 1e3 for i in range(6)]
k = [(c[i] - c[8]) / GHZ / 1e3 for i in range(8, 14)]
print("energies  kernel, workgroup (0,0), us from entry: staging done %.2f | conv done %.2f | query ready %.2f | tanh done %.2f | exit %.2f"
      % (e[1], e[4], e[5], e[2], e[3]))
print("context   kernel, workgroup (0,0), us from entry: e_part summed %.2f | softmax sums %.2f | weights written %.2f | "
      "context partial %.2f | exit %.2f" % (k[1], k[2], k[3], k[4], k[5]))

# attention-cell step kernel of the last frame (stamps of workgroup 0; same s_memtime counter as the attention kernels' stamps)
print("cell step kernel, us from entry: group 0 consumed %.2f | main loop done %.2f | reduced %.2f | exit %.2f ; "
      "cell exit -> energies entry %.2f us ; energies exit -> context entry %.2f us"
      % tuple([(cc[i] - cc[0]) / GHZ / 1e3 for i in (2, 3, 4, 6)] + [(c[0] - cc[6]) / GHZ / 1e3, (c[8] - c[3]) / GHZ / 1e3]))
