"""Flat fp32 parameter / gradient storage with reference-named views.

All learnable parameters of the reference ``Tacotron2`` (state_dict names and shapes) live
in ONE contiguous fp32 device buffer, each tensor 16-byte aligned, so that
  * the optimizer is one fused kernel over the buffer (t2_adam_step) and the gradient norm one reduction,
  * data-parallel training all-reduces ONE buffer over RCCL (112.5 MB for vanilla dims),
  * tensor pairs that the kernels consume as one matrix are adjacent: [mel_out.weight ; gate.weight] is the
    (M+1, D+Ef) projection, the forward/reverse encoder-LSTM input weights are one (8H, E) matrix, etc.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch


def param_manifest(d: dict) -> "OrderedDict[str, tuple]":
    """name -> shape in flat-buffer order (learnable parameters only)."""
    E, k = d["encoded_dim"], d["encoder_kernel_size"]
    Ef = E + (128 if d.get("description_embeddings") else 0)
    P, A, D = d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"]
    M, Pn, Ad = d["num_mels"], d["postnet_dim"], d["att_dim"]
    F, Kl = d.get("loc_filters", 32), d.get("loc_kernel", 31)
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["encoder.embedding.weight"] = (d["num_chars"] + 1, E)
    for i in (0, 4, 8):
        s[f"encoder.convolutions.{i}.weight"] = (E, E, k)
        s[f"encoder.convolutions.{i}.bias"] = (E,)
        s[f"encoder.convolutions.{i + 1}.weight"] = (E,)
        s[f"encoder.convolutions.{i + 1}.bias"] = (E,)
    # adjacent pairs (forward, reverse) -> single (8H, E) / (8H,) operands
    s["encoder.lstm.weight_ih_l0"] = (2 * E, E)
    s["encoder.lstm.weight_ih_l0_reverse"] = (2 * E, E)
    s["encoder.lstm.bias_ih_l0"] = (2 * E,)
    s["encoder.lstm.bias_ih_l0_reverse"] = (2 * E,)
    s["encoder.lstm.bias_hh_l0"] = (2 * E,)
    s["encoder.lstm.bias_hh_l0_reverse"] = (2 * E,)
    s["encoder.lstm.weight_hh_l0"] = (2 * E, E // 2)
    s["encoder.lstm.weight_hh_l0_reverse"] = (2 * E, E // 2)
    if d.get("speaker_tokens"):
        s["speaker_embedding.weight"] = (d["num_speakers"], E)
    if d.get("description_embeddings"):
        s["description_embeddings_linear.0.weight"] = (128, d["description_embeddings_dim"])
        s["description_embeddings_linear.0.bias"] = (128,)
    s["prenet.0.weight"] = (P, M)
    s["prenet.3.weight"] = (P, P)
    s["att_encoder.weight"] = (Ad, Ef)
    s["decoder.att_rnn.weight_ih"] = (4 * A, P + Ef)
    s["decoder.att_rnn.weight_hh"] = (4 * A, A)
    s["decoder.att_rnn.bias_ih"] = (4 * A,)
    s["decoder.att_rnn.bias_hh"] = (4 * A,)
    s["decoder.attention.query_layer.weight"] = (Ad, A)
    s["decoder.attention.v.weight"] = (1, Ad)
    s["decoder.attention.location_conv.weight"] = (F, 2, Kl)
    s["decoder.attention.location_dense.weight"] = (Ad, F)
    s["decoder.lstm.weight_ih"] = (4 * D, A + Ef)
    s["decoder.lstm.weight_hh"] = (4 * D, D)
    s["decoder.lstm.bias_ih"] = (4 * D,)
    s["decoder.lstm.bias_hh"] = (4 * D,)
    # adjacent: (M+1, D+Ef) projection and (M+1,) bias
    s["decoder.mel_out.weight"] = (M, D + Ef)
    s["decoder.gate.weight"] = (1, D + Ef)
    s["decoder.mel_out.bias"] = (M,)
    s["decoder.gate.bias"] = (1,)
    chans = [M, Pn, Pn, Pn, Pn, M]
    for li in range(5):
        s[f"postnet.postnet.{4 * li}.weight"] = (chans[li + 1], chans[li], 5)
        s[f"postnet.postnet.{4 * li + 1}.weight"] = (chans[li + 1],)
        s[f"postnet.postnet.{4 * li + 1}.bias"] = (chans[li + 1],)
    # prosody-controls extension (model/decoder.py:40-48): the reference appends controls_dim input columns to the decoder
    # LSTM's weight_ih and to mel_out.weight.  They are stored as separate blocks so that [att_h | ctx] stays one K segment
    # and [mel_out ; gate] stays one (M+1, D+Ef) matrix; state_dict exchange splits / concatenates (CONTROL_SPLITS).
    C = d.get("controls_dim", 0) if d.get("controls") else 0
    if C:
        s["decoder.lstm.weight_ih#controls"] = (4 * D, C)
        s["decoder.mel_out.weight#controls"] = (M, C)
    return s


CONTROL_SPLITS = ("decoder.lstm.weight_ih", "decoder.mel_out.weight")


# pairs that must be contiguous without padding between them
_GLUED = {
    "encoder.lstm.weight_ih_l0", "encoder.lstm.bias_ih_l0", "encoder.lstm.bias_hh_l0", "encoder.lstm.weight_hh_l0",
    "decoder.mel_out.weight", "decoder.mel_out.bias",
}


def buffer_manifest(d: dict) -> "OrderedDict[str, tuple]":
    """BatchNorm buffers (running stats) - kept outside the learnable flat buffer."""
    E, M, Pn = d["encoded_dim"], d["num_mels"], d["postnet_dim"]
    s: "OrderedDict[str, tuple]" = OrderedDict()
    for i in (1, 5, 9):
        s[f"encoder.convolutions.{i}.running_mean"] = (E,)
        s[f"encoder.convolutions.{i}.running_var"] = (E,)
    chans = [Pn, Pn, Pn, Pn, M]
    for li in range(5):
        s[f"postnet.postnet.{4 * li + 1}.running_mean"] = (chans[li],)
        s[f"postnet.postnet.{4 * li + 1}.running_var"] = (chans[li],)
    return s


def _numel(shape) -> int:
    n = 1
    for x in shape:
        n *= x
    return n


class ParamStore:
    """Owns the flat parameter, gradient and Adam-moment buffers and the name -> view tables."""

    def __init__(self, d: dict, device, with_grad: bool = True):
        self.dims = dict(d)
        self.device = torch.device(device)
        man = param_manifest(d)
        offs, off = OrderedDict(), 0
        for name, shp in man.items():
            offs[name] = off
            n = _numel(shp)
            off += n if name in _GLUED else (n + 3) // 4 * 4
        self.numel = (off + 3) // 4 * 4
        self.offsets, self.shapes = offs, man
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=self.device) if with_grad else None
        self.exp_avg = None
        self.exp_avg_sq = None
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        for name, shp in man.items():
            n = _numel(shp)
            self.P[name] = self.flat[offs[name]:offs[name] + n].view(shp)
            if with_grad:
                self.G[name] = self.grad[offs[name]:offs[name] + n].view(shp)
        bman = buffer_manifest(d)
        nb = sum((_numel(s) + 3) // 4 * 4 for s in bman.values())
        self.buf_flat = torch.zeros(nb, dtype=torch.float32, device=self.device)
        self.Bf: Dict[str, torch.Tensor] = {}
        o = 0
        for name, shp in bman.items():
            n = _numel(shp)
            self.Bf[name] = self.buf_flat[o:o + n].view(shp)
            if name.endswith("running_var"):
                self.Bf[name].fill_(1.0)
            o += (n + 3) // 4 * 4
        self.num_batches_tracked = {k.replace("running_mean", "num_batches_tracked"): 0
                                    for k in bman if k.endswith("running_mean")}

    # --- concatenated operands -------------------------------------------------------------------
    def cat_view(self, first: str, rows: int, cols: int, grad: bool = False) -> torch.Tensor:
        o = self.offsets[first]
        buf = self.grad if grad else self.flat
        return buf[o:o + rows * cols].view(rows, cols) if cols > 0 else buf[o:o + rows]

    def reference_layout(self, table: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
        """Copies of `table` (self.P or self.G) under the reference's names and shapes (controls columns appended)."""
        out = OrderedDict()
        for name, v in table.items():
            if name.endswith("#controls"):
                continue
            c = table.get(name + "#controls")
            out[name] = torch.cat([v, c], 1) if c is not None else v.detach().clone()
        return out

    def init_adam(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)

    # --- state_dict exchange (reference layout) -----------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "", strict: bool = True):
        missing = []
        ctrl = any(n.endswith("#controls") for n in self.P)
        for name in self.P:
            if name.endswith("#controls"):
                continue                     # filled together with its base tensor below
            key = prefix + name
            if key in sd:
                t = torch.as_tensor(sd[key]).to(torch.float32)
                if ctrl and name in CONTROL_SPLITS:       # reference layout: [main columns | controls columns]
                    k0 = self.shapes[name][1]
                    self.P[name + "#controls"].copy_(t[:, k0:])
                    t = t[:, :k0]
                self.P[name].copy_(t.reshape(self.shapes[name]))
            else:
                missing.append(key)
        for name in self.Bf:
            key = prefix + name
            if key in sd:
                self.Bf[name].copy_(torch.as_tensor(sd[key]).to(torch.float32))
            else:
                missing.append(key)
        for name in self.num_batches_tracked:
            if prefix + name in sd:
                self.num_batches_tracked[name] = int(sd[prefix + name])
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        return missing

    def state_dict(self, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        for name, v in self.reference_layout(self.P).items():
            out[prefix + name] = v
        for name, v in self.Bf.items():
            out[prefix + name] = v.detach().clone()
        for name, v in self.num_batches_tracked.items():
            out[prefix + name] = torch.tensor(v, dtype=torch.int64)
        return out
