"""Timing experiments for the LSTM step kernel outside the model (diagnostic, GPU box only).

A: one step repeated (inputs/weights cache-hot)       B: a real recurrence of n steps through t2_lstm_seq_fwd
C: two different cells alternating (weights exceed the 32 MB of L2)
"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import _lib
from tacotron2_amd._lib import call, make

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def pack(W, H):
    K = W.shape[1]
    arr = (_lib.S["T2Seg"] * 1)()
    arr[0].w = W.data_ptr(); arr[0].ldw = K; arr[0].K = K
    ntpad = (K // 16 + 15) // 16 * 16
    out = torch.empty(H // 4 * ntpad * 256, device=dev)
    call("t2_lstm_pack_fwd", arr, 1, H, out, st)
    return out


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def cell(B, H, K, n, ld=None):
    ld = ld or K
    W = torch.randn(4 * H, K, device=dev) / K ** 0.5
    wp = pack(W, H)
    X = torch.randn(n + 1, B, ld, device=dev) * 0.1
    pre = torch.randn(n, B, 4 * H, device=dev)
    c = torch.zeros(n + 1, B, H, device=dev)
    return dict(B=B, H=H, K=K, ld=ld, W=W, wp=wp, X=X, pre=pre, c=c, n=n)


def step_struct(C, t, hout_t, tiled=False):
    B, H, K, ld = C["B"], C["H"], C["K"], C["ld"]
    s = make("T2LstmStep", B=B, H=H, nseg=1, wpacked=C["wp"], pre=C["pre"][t], ldpre=4 * H, c_prev=C["c"][t], ldc_prev=H,
             h_out=C["X"][hout_t], ldh=ld, c_out=C["c"][t + 1], ldc_out=H)
    if tiled:
        Bp = (B + 15) // 16 * 16
        if "Xt" not in C:
            C["Xt"] = torch.randn(C["n"] + 1, K // 16, Bp, 16, device=dev) * 0.1
        s.xt = C["Xt"][t].data_ptr(); s.ht_out = C["Xt"][hout_t].data_ptr(); s.ht_col0 = 0
    s.seg[0].x = C["X"][t].data_ptr(); s.seg[0].ldx = ld; s.seg[0].w = C["W"].data_ptr(); s.seg[0].ldw = K; s.seg[0].K = K
    return s


def main():
    n = 400
    for (B, H, K) in [(32, 1024, 1536), (32, 1024, 1024), (32, 1024, 512), (16, 1024, 1536), (32, 1024, 3072)]:
        C = cell(B, H, K, n)
        sA = step_struct(C, 0, 1)
        tA = timed(lambda: [call("t2_lstm_step_fwd", sA, 1, st) for _ in range(n)], n)
        sB = step_struct(C, 0, 1)
        inc = make("T2LstmStride", pre=B * 4 * H, c_prev=B * H, h_out=B * C["ld"], c_out=B * H, dt=0)
        inc.seg_x[0] = B * C["ld"]
        tB = timed(lambda: call("t2_lstm_seq_fwd", sB, inc, 1, n, st), n)
        C2 = cell(B, H, K, n)
        s1 = step_struct(C, 0, 1); s2 = step_struct(C2, 0, 1)
        def alt():
            for _ in range(n // 2):
                call("t2_lstm_step_fwd", s1, 1, st); call("t2_lstm_step_fwd", s2, 1, st)
        tC = timed(alt, n)
        sT = step_struct(C, 0, 1, tiled=True)
        tT = timed(lambda: [call("t2_lstm_step_fwd", sT, 1, st) for _ in range(n)], n)
        import ctypes
        out8 = (ctypes.c_uint64 * 8)()
        _lib.lib().t2_debug_clock(1, out8)
        call("t2_lstm_step_fwd", sT, 1, st); torch.cuda.synchronize()
        _lib.lib().t2_debug_clock(0, out8)
        ck = [out8[i] for i in range(8)]
        stamps = "in-kernel entry->exit = %d shader cycles, %d x10ns" % (ck[6] - ck[0], ck[7] - ck[1])
        wmb = 4 * H * K * 4 / 1e6
        print(f"B={B} H={H} K={K} weights {wmb:.1f} MB | A same-step {tA:.2f} us | B recurrence {tB:.2f} us | C two cells alternating {tC:.2f} us | tiled same-step {tT:.2f} us | {stamps}", flush=True)


if __name__ == "__main__":
    main()
