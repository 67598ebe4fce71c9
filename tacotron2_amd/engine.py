"""Host-side orchestration of the gfx950 Tacotron 2 hot path (teacher-forced forward / backward, optimizer).

Everything numerical is a call into libtacotron2_amd.so (include/tacotron2_amd.h); torch is used only to own
device memory and the stream.  The forward follows model/tacotron2.py:155-347 of the reference, re-scheduled for
the hardware:

  * every nn.Linear / nn.Conv1d with a full (batch x time) row block is ONE large fp32-MFMA GEMM
    (conv = GEMM over overlapping channel-last rows), including the parts of the two LSTMCells' input
    projections that do not depend on the recurrence (prenet -> att_rnn, [att_h, ctx] -> decoder lstm);
  * in teacher-forced mode the attention recurrence (att_rnn + attention) does not depend on the decoder
    LSTM, so the frame loop runs as two chains: attention chain (3 launches / frame) then decoder-LSTM chain
    (1 launch / frame), and the mel/stop projection becomes one GEMM over all frames.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from ._lib import call, make
from .params import ParamStore

KL, KPAD = 31, 15


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_k=1, b_k=1, alpha=1.0, bias=None, bias2=None, mulmask=None, ldmask=0,
         relu=0, accumulate=0, splitk=1, batch=1, sA=0, sB=0, sC=0):
    """C-ABI t2_gemm on raw pointers (ints) or tensors."""
    g = make("T2Gemm", A=A, B=B, C=C, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, a_kmajor=a_k, b_kmajor=b_k,
             alpha=alpha, bias=bias, bias2=bias2, mulmask=mulmask, ldmask=ldmask, relu=relu, accumulate=accumulate,
             splitk=splitk, batch=batch, sA=sA, sB=sB, sC=sC)
    call("t2_gemm", g, _stream())


def _ptr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + 4 * elem_off


def splitk_for(M: int, N: int, K: int) -> int:
    """Enough K-slices to put ~2 workgroups on each of the 256 CUs for weight-gradient shaped GEMMs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ks = max(1, min((K + 31) // 32, (512 + tiles - 1) // tiles))
    return ks


class Engine:
    """Forward/backward of the whole model on one device.  `ps` is the ParamStore."""

    def __init__(self, ps: ParamStore):
        self.ps = ps
        self.d = ps.dims
        self.dev = ps.device
        self._ws: Dict[str, torch.Tensor] = {}

    # ---- workspace --------------------------------------------------------------------------------
    def buf(self, name: str, *shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        t = self._ws.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            self._ws[name] = t
        v = t[:n].view(*shape)
        if zero:
            v.zero_()
        return v

    # ---- encoder ----------------------------------------------------------------------------------
    def conv_bn_fwd(self, tag, x_pad, w, bias, bn_prefix, B, L, Ci, Co, act, drop, training, ctx,
                    y=None, Lp_y=None, pad_y=2, res=None, Lp_res=0, pad_res=0, length=None, fill=0.0):
        """x_pad (B, L+4, Ci) padded layout -> y (padded layout by default).  Saves raw conv output + stats."""
        P, Bf = self.ps.P, self.ps.Bf
        Lp = L + 4
        wp = self.buf(f"{tag}.wp", Co, 5 * Ci)
        call("t2_pack_conv_weight", w, wp, Co, Ci, 5, 0, _stream())
        raw = self.buf(f"{tag}.raw", B * Lp, Co)
        gemm(x_pad, wp, raw, B * Lp - 4, Co, 5 * Ci, Ci, 5 * Ci, Co, bias=bias)
        mean = self.buf(f"{tag}.mean", Co)
        invstd = self.buf(f"{tag}.invstd", Co)
        sums = self.buf(f"{tag}.sums", 2 * Co, dtype=torch.float64)
        if y is None:
            y = self.buf(f"{tag}.y", B, Lp, Co)
            Lp_y = Lp
        bn = make("T2Bn", B=B, L=L, C=Co, x=raw, Lp_x=Lp, gamma=P[bn_prefix + ".weight"], beta=P[bn_prefix + ".bias"],
                  running_mean=Bf[bn_prefix + ".running_mean"], running_var=Bf[bn_prefix + ".running_var"],
                  training=1 if training else 0, momentum=0.1, eps=1e-5, sums=sums, mean=mean, invstd=invstd, act=act,
                  drop=drop, res=res, Lp_res=Lp_res, pad_res=pad_res, len=length, fill=fill, y=y, Lp_y=Lp_y, pad_y=pad_y)
        call("t2_bn_fwd", bn, _stream())
        if training:
            self.ps.num_batches_tracked[bn_prefix + ".num_batches_tracked"] += 1
        ctx[tag] = dict(x_pad=x_pad, raw=raw, mean=mean, invstd=invstd, drop=drop, wp=wp)
        return y

    def encoder_fwd(self, chars_idx, chars_len32, training, masks, ctx):
        d, P = self.d, self.ps.P
        B, L = chars_idx.shape
        E, H = d["encoded_dim"], d["encoded_dim"] // 2
        Lp = L + 4
        x = self.buf("enc.x0", B, Lp, E)
        call("t2_embedding_fwd", chars_idx, P["encoder.embedding.weight"], x, B, L, E, 2, _stream())
        enc_drop = masks.get("enc_drop") if masks else None
        for li, i in enumerate((0, 4, 8)):
            x = self.conv_bn_fwd(f"enc.conv{li}", x, P[f"encoder.convolutions.{i}.weight"],
                                 P[f"encoder.convolutions.{i}.bias"], f"encoder.convolutions.{i + 1}", B, L, E, E, 1,
                                 enc_drop[li] if enc_drop is not None else None, training, ctx)
        # BiLSTM: one input GEMM for both directions (N = 8H), then L steps with both directions per launch
        pre = self.buf("enc.pre", B * Lp, 8 * H)
        w_ih = self.ps.cat_view("encoder.lstm.weight_ih_l0", 8 * H, E)
        b_ih = self.ps.cat_view("encoder.lstm.bias_ih_l0", 8 * H, 0)
        b_hh = self.ps.cat_view("encoder.lstm.bias_hh_l0", 8 * H, 0)
        gemm(_ptr(x, 2 * E), w_ih, pre, B * Lp - 4, 8 * H, E, E, E, 8 * H, bias=b_ih, bias2=b_hh)
        S = L
        hs = self.buf("enc.h", 2, S + 1, B, H)
        cs = self.buf("enc.c", 2, S + 1, B, H)
        gs = self.buf("enc.gates", 2, S, B, 4 * H)
        hs[0, 0].zero_(); cs[0, 0].zero_(); hs[1, S].zero_(); cs[1, S].zero_()
        enc = self.buf("enc.out", B, L, E)
        steps = (_lib.S["T2LstmStep"] * 2)()
        incs = (_lib.S["T2LstmStride"] * 2)()
        for dr in range(2):
            t0 = 0 if dr == 0 else S - 1
            sg = 1 if dr == 0 else -1
            whh = P["encoder.lstm.weight_hh_l0" + ("" if dr == 0 else "_reverse")]
            slot_in = 0 if dr == 0 else S          # slot holding the previous state
            slot_out = 1 if dr == 0 else S - 1
            st = steps[dr]
            st.B, st.H, st.nseg = B, H, 1
            st.seg[0].x = _ptr(hs[dr, slot_in]); st.seg[0].ldx = H
            st.seg[0].w = whh.data_ptr(); st.seg[0].ldw = H; st.seg[0].K = H
            st.pre = _ptr(pre, t0 * 8 * H + dr * 4 * H); st.ldpre = Lp * 8 * H
            st.c_prev = _ptr(cs[dr, slot_in]); st.ldc_prev = H
            st.h_out = _ptr(hs[dr, slot_out]); st.ldh = H
            st.h_out2 = _ptr(enc, t0 * E + dr * H); st.ldh2 = L * E
            st.c_out = _ptr(cs[dr, slot_out]); st.ldc_out = H
            st.gates_out = _ptr(gs[dr, t0]); st.ldg = 4 * H
            st.len = chars_len32.data_ptr(); st.t = t0
            ic = incs[dr]
            ic.seg_x[0] = sg * B * H
            ic.pre = sg * 8 * H; ic.c_prev = sg * B * H; ic.h_out = sg * B * H; ic.h_out2 = sg * E
            ic.c_out = sg * B * H; ic.gates_out = sg * B * 4 * H; ic.dt = sg
        call("t2_lstm_seq_fwd", steps, incs, 2, S, _stream())
        ctx["enc"] = dict(x3=x, pre=pre, hs=hs, cs=cs, gs=gs, B=B, L=L)
        return enc

    # ---- full teacher-forced forward -------------------------------------------------------------
    def forward_tf(self, chars_idx, chars_len, mel, mel_len, speaker_id=None, description_embeddings=None,
                   training=True, masks: Optional[dict] = None, save_for_backward=True):
        """Returns (mels, mels_post, gates, alignments), ctx.  masks: oracle-convention dict (see oracle.tacotron2_ref)
        already on the device, with prenet/att/dec masks time-major; None entries = identity."""
        d, P, ps = self.d, self.ps.P, self.ps
        masks = masks or {}
        B, L = chars_idx.shape
        T, M = mel.shape[1], d["num_mels"]
        E, Pd, A, D, Ad = d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"], d["att_dim"]
        Ef = E + (128 if d.get("description_embeddings") else 0)
        F = d.get("loc_filters", 32)
        ctx: dict = dict(B=B, L=L, T=T)
        st = _stream()
        len32 = chars_len.to(torch.int32)
        mlen32 = mel_len.to(torch.int32)
        ctx["len32"], ctx["mlen32"], ctx["chars_idx"] = len32, mlen32, chars_idx

        enc = self.encoder_fwd(chars_idx, len32, training, masks, ctx)

        # conditioning (model/tacotron2.py:201-229)
        memory = self.buf("memory", B, L, Ef)
        desc = None
        if d.get("description_embeddings"):
            desc = self.buf("desc", B, 128)
            Dd = d["description_embeddings_dim"]
            gemm(description_embeddings, P["description_embeddings_linear.0.weight"], desc, B, 128, Dd, Dd, Dd, 128)
            call("t2_tanh_bias", desc, P["description_embeddings_linear.0.bias"], B, 128, st)
        spk32 = speaker_id.to(torch.int32) if d.get("speaker_tokens") else None
        call("t2_condition_fwd", enc, P["speaker_embedding.weight"] if d.get("speaker_tokens") else None, spk32, desc,
             memory, B, L, E, Ef, st)
        ctx.update(enc=enc, memory=memory, desc=desc, spk32=spk32, desc_in=description_embeddings)
        pmT = self.buf("pmT", B, Ad, L)   # processed memory, transposed: one batched GEMM W_att x memory[b]^T
        gemm(P["att_encoder.weight"], memory, pmT, Ad, L, Ef, Ef, Ef, L, batch=B, sA=0, sB=L * Ef, sC=Ad * L)

        # prenet on all frames, time-major (model/tacotron2.py:255-258)
        mel_tm = self.buf("mel_tm", T + 1, B, M)
        call("t2_mel_to_tm", mel, mel_tm, B, T, M, st)
        pd = masks.get("prenet_drop")
        p1 = self.buf("p1", T + 1, B, Pd)
        p2 = self.buf("p2", T + 1, B, Pd)
        R1 = (T + 1) * B
        gemm(mel_tm, P["prenet.0.weight"], p1, R1, Pd, M, M, M, Pd, relu=1, mulmask=pd[0] if pd else None, ldmask=Pd)
        gemm(p1, P["prenet.3.weight"], p2, R1, Pd, Pd, Pd, Pd, Pd, relu=1, mulmask=pd[1] if pd else None, ldmask=Pd)

        # hoisted prenet part of the attention-RNN input projection + both biases
        R = T * B
        pre_att = self.buf("pre_att", T, B, 4 * A)
        gemm(p2, P["decoder.att_rnn.weight_ih"], pre_att, R, 4 * A, Pd, Pd, Pd + Ef, 4 * A,
             bias=P["decoder.att_rnn.bias_ih"], bias2=P["decoder.att_rnn.bias_hh"])

        # attention chain
        U = self.buf("U", Ad, 2, KL)
        call("t2_attn_fold_location", P["decoder.attention.location_dense.weight"],
             P["decoder.attention.location_conv.weight"], U, Ad, F, KL, st)
        xdec = self.buf("xdec", T + 1, B, A + Ef)
        xdec[0].zero_()
        att_c = self.buf("att_c", T + 1, B, A)
        att_c[0].zero_()
        cum = self.buf("cum", T + 1, B, L)
        cum[0].zero_()
        xproj = self.buf("xproj", T + 1, B, D + Ef)
        xproj[0].zero_()
        gates_att = self.buf("gates_att", T, B, 4 * A) if save_for_backward else None
        th = self.buf("th", T, B, Ad, L) if save_for_backward else None
        align = torch.empty(B, T, L, dtype=torch.float32, device=self.dev)
        e_part = self.buf("e_part", B, Ad // 16, L)
        seq = make("T2AttnSeq", B=B, L=L, T=T, A=A, Ad=Ad, Ef=Ef, Kl=KL,
                   W_ih_ctx=_ptr(P["decoder.att_rnn.weight_ih"], Pd), ld_wih=Pd + Ef,
                   W_hh=P["decoder.att_rnn.weight_hh"], Wq=P["decoder.attention.query_layer.weight"], U=U,
                   v=P["decoder.attention.v.weight"], pre=pre_att, pmT=pmT, memory=memory, len=len32,
                   att_drop=masks.get("att_drop"), xdec=xdec, att_c=att_c, gates=gates_att, align=align, cum=cum, th=th,
                   xproj_ctx=_ptr(xproj, B * (D + Ef) + D), ld_xproj=D + Ef, e_part=e_part)
        call("t2_attn_seq_fwd", seq, st)

        # decoder-LSTM chain: hoisted input projection, then T recurrent steps
        pre_dec = self.buf("pre_dec", T, B, 4 * D)
        gemm(_ptr(xdec, B * (A + Ef)), P["decoder.lstm.weight_ih"], pre_dec, R, 4 * D, A + Ef, A + Ef, A + Ef, 4 * D,
             bias=P["decoder.lstm.bias_ih"], bias2=P["decoder.lstm.bias_hh"])
        dec_c = self.buf("dec_c", T + 1, B, D)
        dec_c[0].zero_()
        gates_dec = self.buf("gates_dec", T, B, 4 * D) if save_for_backward else None
        dd = masks.get("dec_drop")
        ldp = D + Ef
        stp = make("T2LstmStep", B=B, H=D, nseg=1, pre=pre_dec, ldpre=4 * D, c_prev=dec_c, ldc_prev=D,
                   drop=dd, lddrop=D, h_out=_ptr(xproj, B * ldp), ldh=ldp, c_out=_ptr(dec_c, B * D), ldc_out=D,
                   gates_out=gates_dec, ldg=4 * D)
        stp.seg[0].x = xproj.data_ptr(); stp.seg[0].ldx = ldp
        stp.seg[0].w = P["decoder.lstm.weight_hh"].data_ptr(); stp.seg[0].ldw = D; stp.seg[0].K = D
        inc = make("T2LstmStride", pre=B * 4 * D, c_prev=B * D, drop=B * D, h_out=B * ldp, c_out=B * D,
                   gates_out=B * 4 * D, dt=0)
        inc.seg_x[0] = B * ldp
        call("t2_lstm_seq_fwd", stp, inc, 1, T, st)

        # mel + stop projection over all frames: [mel_out.weight ; gate.weight] is one (M+1, D+Ef) matrix
        wproj = ps.cat_view("decoder.mel_out.weight", M + 1, D + Ef)
        bproj = ps.cat_view("decoder.mel_out.bias", M + 1, 0)
        proj = self.buf("proj", T, B, M + 1)
        gemm(_ptr(xproj, B * ldp), wproj, proj, R, M + 1, ldp, ldp, ldp, M + 1, bias=bproj)
        mels = torch.empty(B, T, M, dtype=torch.float32, device=self.dev)
        gates = torch.empty(B, T, 1, dtype=torch.float32, device=self.dev)
        post_in = self.buf("post.x0", B, T + 4, M)
        call("t2_finalize_fwd", proj, mlen32, mels, gates, post_in, B, T, M, st)

        # postnet (model/postnet.py) + residual + output masks (model/tacotron2.py:331-345)
        Pn = d["postnet_dim"]
        chans = [M, Pn, Pn, Pn, Pn, M]
        post_drop = masks.get("post_drop")
        x = post_in
        post = torch.empty(B, T, M, dtype=torch.float32, device=self.dev)
        for li in range(5):
            last = li == 4
            x = self.conv_bn_fwd(f"post.conv{li}", x, P[f"postnet.postnet.{4 * li}.weight"], None,
                                 f"postnet.postnet.{4 * li + 1}", B, T, chans[li], chans[li + 1], 0 if last else 2,
                                 post_drop[li] if post_drop is not None else None, training, ctx,
                                 y=post if last else None, Lp_y=T if last else None, pad_y=0 if last else 2,
                                 res=post_in if last else None, Lp_res=T + 4, pad_res=2,
                                 length=mlen32 if last else None, fill=0.0)
        ctx.update(pmT=pmT, mel_tm=mel_tm, p1=p1, p2=p2, pd=pd, pre_att=pre_att, U=U, xdec=xdec, att_c=att_c, cum=cum,
                   xproj=xproj, gates_att=gates_att, th=th, align=align, pre_dec=pre_dec, dec_c=dec_c,
                   gates_dec=gates_dec, proj=proj, post_in=post_in, masks=masks, training=training)
        return (mels, post, gates, align), ctx
