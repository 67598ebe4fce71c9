"""Encoder / Attention / Decoder / Postnet with the reference's call signatures (model/encoder.py:54, model/attention.py:34-41,
model/decoder.py:53-67, model/postnet.py:51).  In the reference these are only called from Tacotron2.forward; they are
kept so code written against the sub-modules still runs.  They execute the same HIP kernels as the fused path through
the owning Tacotron2's engine (inference-style: no autograd through sub-module calls; train through Tacotron2.forward)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from .._lib import call, make
from ..engine import KL, _ptr, _stream


class _Sub(nn.Module):
    def _root(self):
        return self.__dict__["_t2_root"]()

    def _P(self):
        return self._root().store.P


class Encoder(_Sub):
    def forward(self, char_idx: Tensor, char_idx_len: Tensor) -> Tensor:
        root = self._root()
        with torch.no_grad():
            ctx: dict = {}
            enc = root._engine.encoder_fwd(char_idx.contiguous(), char_idx_len.to(torch.int32), root.training, {}, ctx)
            return enc[:, :int(char_idx_len.max())].clone()        # pad_packed_sequence length (model/encoder.py:65)


class Attention(_Sub):
    def forward(self, attention_hidden_state: Tensor, memory: Tensor, processed_memory: Tensor,
                attention_weights_cat: Tensor, mask: Tensor):
        P = self._P()
        B, L, Ef = memory.shape
        Wq = P["decoder.attention.query_layer.weight"]
        Ad, A = Wq.shape
        dev = memory.device
        with torch.no_grad():
            U = torch.empty(Ad, 2, KL, device=dev)
            call("t2_attn_fold_location", P["decoder.attention.location_dense.weight"],
                 P["decoder.attention.location_conv.weight"], U, Ad, P["decoder.attention.location_dense.weight"].shape[1], KL,
                 _stream())
            pmT = processed_memory.transpose(1, 2).contiguous()
            lens = (L - mask.sum(1)).to(torch.int32)                # suffix mask (model/tacotron2.py:223-226)
            wcat = attention_weights_cat.contiguous()
            w = torch.empty(B, L, device=dev); ctxv = torch.empty(B, Ef, device=dev)
            ws = torch.empty(B, Ad // 16, L, device=dev)
            s = make("T2AttnStep", B=B, L=L, A=A, Ad=Ad, Ef=Ef, Kl=KL, att_h=attention_hidden_state.contiguous(), ldh=A,
                     Wq=Wq, U=U, v=P["decoder.attention.v.weight"], w_prev=_ptr(wcat, 0), ldw=2 * L,
                     cum_prev=_ptr(wcat, L), ldcum=2 * L, pmT=pmT, memory=memory.contiguous(), len=lens, e_part=ws,
                     w_out=w, ldwo=L, ctx_out=ctxv, ldctx=Ef)
            s._keep += [wcat]
            call("t2_attn_step_fwd", s, _stream())
        return ctxv, w


class Decoder(_Sub):
    def forward(self, prev_mel_prenet: Tensor, att_rnn_hidden: Tuple[Tensor, Tensor], att_context, att_weights,
                att_weights_cum, rnn_hidden: Tuple[Tensor, Tensor], encoded, att_encoded, encoded_mask,
                speech_features: Optional[Tensor] = None, extra_att_in: Optional[Tensor] = None,
                extra_decoder_in: Optional[Tensor] = None):
        assert speech_features is None and extra_att_in is None, \
            "speech_features / extra_att_in are never passed by the reference's Tacotron2 (model/tacotron2.py:288-301)"
        root = self._root()
        P = self._P()
        dev = encoded.device
        B = encoded.shape[0]
        A, D = root.att_rnn_dim, root.rnn_hidden_dim
        Pd = prev_mel_prenet.shape[1]
        Ef = encoded.shape[2]
        M = root.num_mels
        training = root.training
        with torch.no_grad():
            st = _stream()
            g = torch.Generator(device="cpu")
            def drop(n):
                if not training:
                    return None
                m = torch.empty(B, n, device=dev)
                call("t2_philox_mask", m, B * n, 0.1, int(torch.randint(0, 2 ** 31, (1,))), 0, st)
                return m
            # attention LSTM cell: [prenet | context] and recurrent part as three segments of the generic step kernel
            xin = [prev_mel_prenet.contiguous(), att_context.contiguous(), att_rnn_hidden[0].contiguous()]
            att_h = torch.empty(B, A, device=dev); att_c = torch.empty(B, A, device=dev)
            dm = drop(A)
            s = make("T2LstmStep", B=B, H=A, nseg=3, bias1=P["decoder.att_rnn.bias_ih"], bias2=P["decoder.att_rnn.bias_hh"],
                     c_prev=att_rnn_hidden[1].contiguous(), ldc_prev=A, drop=dm, lddrop=A, h_out=att_h, ldh=A, c_out=att_c,
                     ldc_out=A)
            Wih = P["decoder.att_rnn.weight_ih"]
            for i, (x, w, ld, K) in enumerate(((xin[0], _ptr(Wih, 0), Pd + Ef, Pd), (xin[1], _ptr(Wih, Pd), Pd + Ef, Ef),
                                               (xin[2], P["decoder.att_rnn.weight_hh"].data_ptr(), A, A))):
                s.seg[i].x = x.data_ptr(); s.seg[i].ldx = x.shape[1]; s.seg[i].w = w; s.seg[i].ldw = ld; s.seg[i].K = K
            s._keep += xin
            call("t2_lstm_step_fwd", s, 1, st)
            ctxv, w = self._modules["attention"](att_h, encoded, att_encoded,
                                                torch.stack([att_weights, att_weights_cum], 1), encoded_mask)
            att_weights_cum += w                                   # in place, like model/decoder.py:90
            xin2 = [att_h, ctxv, rnn_hidden[0].contiguous()]
            rnn_h = torch.empty(B, D, device=dev); rnn_c = torch.empty(B, D, device=dev)
            dm2 = drop(D)
            d = make("T2LstmStep", B=B, H=D, nseg=3, bias1=P["decoder.lstm.bias_ih"], bias2=P["decoder.lstm.bias_hh"],
                     c_prev=rnn_hidden[1].contiguous(), ldc_prev=D, drop=dm2, lddrop=D, h_out=rnn_h, ldh=D, c_out=rnn_c,
                     ldc_out=D)
            Wd = P["decoder.lstm.weight_ih"]
            for i, (x, w_, ld, K) in enumerate(((xin2[0], _ptr(Wd, 0), A + Ef, A), (xin2[1], _ptr(Wd, A), A + Ef, Ef),
                                                (xin2[2], P["decoder.lstm.weight_hh"].data_ptr(), D, D))):
                d.seg[i].x = x.data_ptr(); d.seg[i].ldx = x.shape[1]; d.seg[i].w = w_; d.seg[i].ldw = ld; d.seg[i].K = K
            d._keep += xin2
            cterm = cmel1 = None
            if extra_decoder_in is not None:       # the controls columns of weight_ih / mel_out.weight (model/decoder.py:94-109)
                _, cterm, cmel1 = root._engine.controls_terms(extra_decoder_in, B)
                d.pre = cterm.data_ptr(); d.ldpre = 4 * D
            call("t2_lstm_step_fwd", d, 1, st)
            hc = torch.cat([rnn_h, ctxv], 1).contiguous()
            out = torch.empty(B, M + 1, device=dev)
            wproj = root.store.cat_view("decoder.mel_out.weight", M + 1, D + Ef)
            bproj = root.store.cat_view("decoder.mel_out.bias", M + 1, 0)
            call("t2_linear_rows", hc, D + Ef, wproj, D + Ef, bproj, None, 0, 0, out, M + 1, B, M + 1, D + Ef, st)
            if cmel1 is not None:
                out += cmel1
        return (out[:, :M].contiguous(), out[:, M:].contiguous(), (att_h, att_c), ctxv, w, att_weights_cum, (rnn_h, rnn_c))


class Postnet(_Sub):
    def forward(self, X: Tensor) -> Tensor:
        """X (B, num_mels, T) channel-first like the reference; returns the postnet residual (B, num_mels, T)."""
        root = self._root()
        P = self._P()
        eng = root._engine
        d = root.dims
        B, M, T = X.shape
        Pn = d["postnet_dim"]
        with torch.no_grad():
            x = torch.zeros(B, T + 4, M, device=X.device)
            x[:, 2:T + 2] = X.transpose(1, 2)
            chans = [M, Pn, Pn, Pn, Pn, M]
            ctx: dict = {}
            out = torch.empty(B, T, M, device=X.device)
            p = float(d["dropout"])
            for li in range(5):
                last = li == 4
                dm = None
                if root.training and p > 0:
                    dm = torch.empty(B, T, chans[li + 1], device=X.device)
                    call("t2_philox_mask", dm, dm.numel(), p, int(torch.randint(0, 2 ** 31, (1,))), li, _stream())
                x = eng.conv_bn_fwd(f"sub.post{li}", x, P[f"postnet.postnet.{4 * li}.weight"], None,
                                    f"postnet.postnet.{4 * li + 1}", B, T, chans[li], chans[li + 1], 0 if last else 2, dm,
                                    root.training, ctx, y=out if last else None, Lp_y=T if last else None,
                                    pad_y=0 if last else 2)
        return out.transpose(1, 2).contiguous()


CLASSES = {"encoder": Encoder, "decoder": Decoder, "decoder.attention": Attention, "postnet": Postnet}
