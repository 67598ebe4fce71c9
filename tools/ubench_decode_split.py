"""Would split LSTM cells on a second stream shorten the autoregressive decode frame?  (GPU box; TIMING emulation with the product's
kernels - the numbers it computes are meaningless, the launch pattern, operand sizes and dependencies are those of the real frame.)

Product frame (main stream, 6 launches): combined linear (K = 1536, N = 337), second prenet layer (K = 256), attention cell
(K = 1792), energies + context (t2_attn_step_fwd: 2 launches), decoder cell (K = 2560).
Split variant: the parts of the two cells' reductions that do not depend on the frame's own critical inputs - attention cell:
[att_h(t-1) | ctx(t-1)] (K = 1536); decoder cell: dec_h(t-1) (K = 1024) - run as ONE launch of two partial cells on a SECOND stream
behind the previous frame's decoder cell, next to this frame's two small linears; the main stream keeps a K = 256 attention cell and
a K = 1536 decoder cell that add the partial sums (`pre`).  2 event records + 2 stream waits per frame.
Both variants are enqueued behind a 40 ms idle wave so that the host is out of the picture; HIP events give the GPU time per frame.
usage: python tools/ubench_decode_split.py [B ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tacotron2_amd  # noqa
from tacotron2_amd import _lib
from tacotron2_amd._lib import call, make

dev = torch.device("cuda:0")
P, A, Ef, D, Ad, M, L, KL = 256, 1024, 512, 1024, 128, 80, 167, 31
H = 1024


def pack(K, st):
    W = torch.randn(4 * H, K, device=dev) / K ** 0.5
    arr = (_lib.S["T2Seg"] * 1)()
    arr[0].w = W.data_ptr(); arr[0].ldw = K; arr[0].K = K
    ntpad = (K // 16 + 15) // 16 * 16
    out = torch.empty(H // 4 * ntpad * 256, device=dev)
    call("t2_lstm_pack_fwd", arr, 1, H, out, st)
    return out, W


def cell(B, K, st):
    Bp = (B + 15) // 16 * 16
    wp, W = pack(K, st)
    d = dict(wp=wp, W=W, xt=torch.randn(K // 16, Bp, 16, device=dev) * 0.1, pre=torch.randn(B, 4 * H, device=dev) * 0.1,
             c=torch.zeros(2, B, H, device=dev), h=torch.empty(B, H, device=dev), ht=torch.empty(H // 16, Bp, 16, device=dev),
             x=torch.randn(B, K, device=dev) * 0.1)
    s = make("T2LstmStep", B=B, H=H, nseg=1, wpacked=d["wp"], pre=d["pre"], ldpre=4 * H, c_prev=d["c"][0], ldc_prev=H, h_out=d["h"],
             ldh=H, c_out=d["c"][1], ldc_out=H, xt=d["xt"], ht_out=d["ht"], ht_col0=0)
    s.seg[0].x = d["x"].data_ptr(); s.seg[0].ldx = K; s.seg[0].w = W.data_ptr(); s.seg[0].ldw = K; s.seg[0].K = K
    d["s"] = s
    return d


def attn(B):
    t = lambda *s: torch.randn(*s, device=dev) * 0.1
    d = dict(att_h=t(B, A), Wq=t(Ad, A), U=t(Ad, 2, KL), v=t(1, Ad), w=torch.softmax(t(B, L), 1), cum=torch.softmax(t(B, L), 1),
             pmT=t(B, Ad, L), mem=t(B, L, Ef), len=torch.full((B,), L, dtype=torch.int32, device=dev), e=t(B, Ad // 16, L),
             wo=t(B, L), co=t(B, L), ctx=t(B, Ef))
    d["s"] = make("T2AttnStep", B=B, L=L, A=A, Ad=Ad, Ef=Ef, Kl=KL, att_h=d["att_h"], ldh=A, Wq=d["Wq"], U=d["U"], v=d["v"], w_prev=d["w"],
                  ldw=L, cum_prev=d["cum"], ldcum=L, pmT=d["pmT"], memory=d["mem"], len=d["len"], e_part=d["e"], w_out=d["wo"], ldwo=L,
                  cum_out=d["co"], ldco=L, ctx_out=d["ctx"], ldctx=Ef)
    return d


def run(B, frames=150):
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    st, ss = main.cuda_stream, side.cuda_stream
    x1, w1, o1 = torch.randn(B, D + Ef, device=dev), torch.randn(P + M + 1, D + Ef, device=dev) * 0.02, torch.empty(B, P + M + 1, device=dev)
    x2, w2, o2 = torch.randn(B, P, device=dev), torch.randn(P, P, device=dev) * 0.05, torch.empty(B, P, device=dev)
    cA, cD = cell(B, P + A + Ef, st), cell(B, A + Ef + D, st)              # product cells
    cA1, cD1 = cell(B, P, st), cell(B, A + Ef, st)                        # split: what stays on the critical path
    pA, pD = cell(B, A + Ef, st), cell(B, D, st)                          # split: partial cells on the side stream
    at = attn(B)
    two = (_lib.S["T2LstmStep"] * 2)()
    import ctypes
    ctypes.memmove(ctypes.addressof(two[0]), ctypes.addressof(pA["s"]), ctypes.sizeof(pA["s"]))
    ctypes.memmove(ctypes.addressof(two[1]), ctypes.addressof(pD["s"]), ctypes.sizeof(pD["s"]))
    spin = torch.zeros(8, dtype=torch.int32, device=dev)

    def lin():
        call("t2_linear_rows", x1, D + Ef, w1, D + Ef, None, None, 0, 1, o1, P + M + 1, B, P + M + 1, D + Ef, st)
        call("t2_linear_rows", x2, P, w2, P, None, None, 0, 1, o2, P, B, P, P, st)

    def product():
        for _ in range(frames):
            lin()
            call("t2_lstm_step_fwd", cA["s"], 1, st)
            call("t2_attn_step_fwd", at["s"], st)
            call("t2_lstm_step_fwd", cD["s"], 1, st)

    def split(one_launch=True):
        evD = torch.cuda.Event(); evD.record(main)
        for _ in range(frames):
            side.wait_event(evD)
            if one_launch:
                call("t2_lstm_step_fwd", two, 2, ss)
            else:
                call("t2_lstm_step_fwd", pA["s"], 1, ss); call("t2_lstm_step_fwd", pD["s"], 1, ss)
            evP = torch.cuda.Event(); evP.record(side)
            lin()
            main.wait_event(evP)
            call("t2_lstm_step_fwd", cA1["s"], 1, st)
            call("t2_attn_step_fwd", at["s"], st)
            call("t2_lstm_step_fwd", cD1["s"], 1, st)
            evD = torch.cuda.Event(); evD.record(main)

    def timed(fn):
        fn_small = lambda: None
        torch.cuda.synchronize()
        call("t2_stream_probe_spin", spin, 45000, st)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main)
        side.wait_event(e0)
        fn()
        main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / frames

    product(); split(); torch.cuda.synchronize()          # warm-up
    for rep in range(2):
        a = timed(product)
        b = timed(lambda: split(True))
        c = timed(lambda: split(False))
        print(f"B={B:3d}  product frame {a:6.2f} us | split cells, partials as ONE side-stream launch {b:6.2f} us | as two launches {c:6.2f} us", flush=True)


if __name__ == "__main__":
    for B in ([int(x) for x in sys.argv[1:]] or [64, 32, 1]):
        run(B)
