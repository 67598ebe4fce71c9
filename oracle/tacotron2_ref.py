"""CPU restatement of the reference Tacotron 2 hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Plain tensor arithmetic on CPU (matmul / elementwise / explicit loops), written from the
reference's behaviour and NOT calling torch.nn layers, so it is an independent restatement that
autograd can differentiate.  Every stochastic site takes an explicit *scale mask*
(values 0 or 1/(1-p)); ``None`` means "no dropout at this site".

Parameter names and shapes are exactly ``Tacotron2.state_dict()`` of the reference
(SURVEY.md Appendix A).  Activations here are channel-last: (B, L, C) / (B, T, C).

Reference lines followed (paths relative to /root/reference):
  encoder          model/encoder.py:54-67
  conditioning     model/tacotron2.py:197-229
  prenet           model/tacotron2.py:85-92, model/modules.py:4-12
  attention        model/attention.py:34-69
  decoder step     model/decoder.py:53-119
  loop + stop      model/tacotron2.py:231-329
  postnet + masks  model/postnet.py:5-52, model/tacotron2.py:331-347
  loss             model/tts_model.py:197-201
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# dims / parameter manifest
# --------------------------------------------------------------------------------------------
def default_dims(**over) -> dict:
    """Vanilla dims (config/vanilla-ljspeech-stop.json:33-42, run/train.py:218-219)."""
    d = dict(
        num_chars=39, encoded_dim=512, encoder_kernel_size=5, num_mels=80, prenet_dim=256,
        att_rnn_dim=1024, att_dim=128, rnn_hidden_dim=1024, postnet_dim=512, dropout=0.5,
        speaker_tokens=False, num_speakers=1, description_embeddings=False,
        description_embeddings_dim=0, loc_filters=32, loc_kernel=31, controls=False, controls_dim=0,
    )
    d.update(over)
    return d


def enc_full_dim(d: dict) -> int:
    return d["encoded_dim"] + (128 if d["description_embeddings"] else 0)


def param_shapes(d: dict) -> Dict[str, tuple]:
    """name -> shape, the reference's state_dict layout."""
    E, k = d["encoded_dim"], d["encoder_kernel_size"]
    Ef, P, A, D = enc_full_dim(d), d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"]
    M, Pn, Ad = d["num_mels"], d["postnet_dim"], d["att_dim"]
    s: Dict[str, tuple] = {}
    s["encoder.embedding.weight"] = (d["num_chars"] + 1, E)
    for i in (0, 4, 8):
        s[f"encoder.convolutions.{i}.weight"] = (E, E, k)
        s[f"encoder.convolutions.{i}.bias"] = (E,)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"encoder.convolutions.{i + 1}.{n}"] = (E,)
        s[f"encoder.convolutions.{i + 1}.num_batches_tracked"] = ()
    for sfx in ("", "_reverse"):
        s[f"encoder.lstm.weight_ih_l0{sfx}"] = (2 * E, E)
        s[f"encoder.lstm.weight_hh_l0{sfx}"] = (2 * E, E // 2)
        s[f"encoder.lstm.bias_ih_l0{sfx}"] = (2 * E,)
        s[f"encoder.lstm.bias_hh_l0{sfx}"] = (2 * E,)
    if d["speaker_tokens"]:
        s["speaker_embedding.weight"] = (d["num_speakers"], E)
    if d["description_embeddings"]:
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
    s["decoder.attention.location_conv.weight"] = (d["loc_filters"], 2, d["loc_kernel"])
    s["decoder.attention.location_dense.weight"] = (Ad, d["loc_filters"])
    C = d.get("controls_dim", 0) if d.get("controls") else 0      # model/tacotron2.py:119 extra_decoder_in_dim
    s["decoder.lstm.weight_ih"] = (4 * D, A + Ef + C)
    s["decoder.lstm.weight_hh"] = (4 * D, D)
    s["decoder.lstm.bias_ih"] = (4 * D,)
    s["decoder.lstm.bias_hh"] = (4 * D,)
    s["decoder.mel_out.weight"] = (M, D + Ef + C)
    s["decoder.mel_out.bias"] = (M,)
    s["decoder.gate.weight"] = (1, D + Ef)
    s["decoder.gate.bias"] = (1,)
    chans = [M, Pn, Pn, Pn, Pn, M]
    for li in range(5):
        s[f"postnet.postnet.{4 * li}.weight"] = (chans[li + 1], chans[li], 5)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"postnet.postnet.{4 * li + 1}.{n}"] = (chans[li + 1],)
        s[f"postnet.postnet.{4 * li + 1}.num_batches_tracked"] = ()
    return s


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def init_params(d: dict, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random-init parameters with PyTorch-default-like scales (uniform +-1/sqrt(fan_in));
    embeddings N(0, 0.5) as model/encoder.py:26, model/tacotron2.py:65.  BN running stats are
    randomised to non-trivial values so eval-mode tests exercise them."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shp in param_shapes(d).items():
        if name.endswith("num_batches_tracked"):
            P[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            P[name] = (torch.rand(shp, generator=g, dtype=torch.float64) * 0.2 - 0.1).to(dtype)
        elif name.endswith("running_var"):
            P[name] = (torch.rand(shp, generator=g, dtype=torch.float64) * 0.5 + 0.75).to(dtype)
        elif "embedding.weight" in name:
            P[name] = (torch.randn(shp, generator=g, dtype=torch.float64) * 0.5).to(dtype)
            if name == "encoder.embedding.weight":
                P[name][0].zero_()  # padding_idx=0 row (model/encoder.py:25)
        elif len(shp) == 1 and (".convolutions." in name or ".postnet." in name) and name.endswith("weight"):
            P[name] = (torch.rand(shp, generator=g, dtype=torch.float64) * 0.5 + 0.75).to(dtype)  # BN gamma
        elif len(shp) == 1 and (".convolutions." in name or ".postnet." in name) and name.endswith("bias") \
                and int(name.split(".")[2]) % 4 == 1:
            P[name] = (torch.rand(shp, generator=g, dtype=torch.float64) * 0.2 - 0.1).to(dtype)  # BN beta
        else:
            if "lstm" in name or "att_rnn" in name:
                hid = shp[0] // 4
                bound = 1.0 / math.sqrt(hid)
            elif len(shp) >= 2:
                fan_in = 1
                for x in shp[1:]:
                    fan_in *= x
                bound = 1.0 / math.sqrt(fan_in)
            else:
                bound = 0.05
            P[name] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return P


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _sigmoid(x: Tensor) -> Tensor:
    return 1.0 / (1.0 + torch.exp(-x))


def lstm_cell(gates: Tensor, c: Tensor):
    """PyTorch gate order i,f,g,o (model/decoder.py:26-28, nn.LSTMCell)."""
    H = c.shape[-1]
    i = _sigmoid(gates[..., 0 * H:1 * H])
    f = _sigmoid(gates[..., 1 * H:2 * H])
    g = torch.tanh(gates[..., 2 * H:3 * H])
    o = _sigmoid(gates[..., 3 * H:4 * H])
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


def conv1d_cl(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    """'same' 1-D convolution, channel-last.  x (B,L,Cin), w (Cout,Cin,K) -> (B,L,Cout)."""
    B, L, Cin = x.shape
    K = w.shape[2]
    pad = (K - 1) // 2
    xp = torch.zeros(B, L + 2 * pad, Cin, dtype=x.dtype)
    xp[:, pad:pad + L] = x
    out = torch.zeros(B, L, w.shape[0], dtype=x.dtype)
    for k in range(K):
        out = out + xp[:, k:k + L] @ w[:, :, k].T
    if bias is not None:
        out = out + bias
    return out


def batch_norm_cl(x: Tensor, gamma, beta, rmean, rvar, training: bool, new_stats: Optional[dict], key: str):
    """BatchNorm1d over all (B, L) positions of a channel-last tensor, padded positions included
    (SURVEY.md Appendix C.6).  training -> biased batch variance for normalisation; running stats
    updated with the unbiased variance, momentum 0.1 (PyTorch defaults used by the reference)."""
    C = x.shape[-1]
    if training:
        flat = x.reshape(-1, C)
        n = flat.shape[0]
        mean = flat.mean(0)
        var = ((flat - mean) ** 2).mean(0)
        if new_stats is not None:
            unb = var * (n / max(n - 1, 1))
            new_stats[key + ".running_mean"] = ((1 - BN_MOMENTUM) * rmean + BN_MOMENTUM * mean).detach()
            new_stats[key + ".running_var"] = ((1 - BN_MOMENTUM) * rvar + BN_MOMENTUM * unb).detach()
    else:
        mean, var = rmean, rvar
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def encoder_fwd(P, chars_idx: Tensor, chars_len: Tensor, training: bool,
                enc_drop: Optional[List[Tensor]] = None, new_stats: Optional[dict] = None,
                return_conv: bool = False):
    """model/encoder.py:54-67.  Returns encoded (B, max(len), E), zeros past each length."""
    ew = P["encoder.embedding.weight"]
    ew = torch.cat([ew[:1].detach(), ew[1:]], 0)      # padding_idx=0: row 0 receives no gradient
    x = ew[chars_idx]                                                  # (B,L,E)
    for li, i in enumerate((0, 4, 8)):
        x = conv1d_cl(x, P[f"encoder.convolutions.{i}.weight"], P[f"encoder.convolutions.{i}.bias"])
        x = batch_norm_cl(x, P[f"encoder.convolutions.{i + 1}.weight"], P[f"encoder.convolutions.{i + 1}.bias"],
                          P[f"encoder.convolutions.{i + 1}.running_mean"],
                          P[f"encoder.convolutions.{i + 1}.running_var"], training, new_stats,
                          f"encoder.convolutions.{i + 1}")
        x = torch.relu(x)
        if enc_drop is not None and enc_drop[li] is not None:
            x = x * enc_drop[li]
    conv_out = x
    B, L, E = x.shape
    H = E // 2
    # pad_packed_sequence returns max(len) columns; batches are padded to their own longest item so this equals L
    # (Appendix C.7).  Data-parallel shards padded to the GLOBAL L keep L columns (zeros past each length).
    Lmax = L
    out = torch.zeros(B, Lmax, E, dtype=x.dtype)
    lens = chars_len.to(torch.int64)
    for d_, sfx in enumerate(("", "_reverse")):
        Wih, Whh = P[f"encoder.lstm.weight_ih_l0{sfx}"], P[f"encoder.lstm.weight_hh_l0{sfx}"]
        b = P[f"encoder.lstm.bias_ih_l0{sfx}"] + P[f"encoder.lstm.bias_hh_l0{sfx}"]
        h = torch.zeros(B, H, dtype=x.dtype)
        c = torch.zeros(B, H, dtype=x.dtype)
        steps = range(Lmax) if d_ == 0 else range(Lmax - 1, -1, -1)
        outs = [None] * Lmax
        for t in steps:
            gates = x[:, t] @ Wih.T + h @ Whh.T + b
            hn, cn = lstm_cell(gates, c)
            act = (t < lens)[:, None]
            h = torch.where(act, hn, h)
            c = torch.where(act, cn, c)
            outs[t] = torch.where(act, hn, torch.zeros_like(hn))
        out[:, :, d_ * H:(d_ + 1) * H] = torch.stack(outs, 1)
    if return_conv:
        return out, conv_out
    return out


def condition(P, d: dict, encoded: Tensor, speaker_id=None, description_embeddings=None):
    """model/tacotron2.py:201-229 -> (encoded_full, att_encoded)."""
    if d["speaker_tokens"]:
        encoded = torch.tanh(encoded + P["speaker_embedding.weight"][speaker_id.long()][:, None, :])
    if d["description_embeddings"] and description_embeddings is not None:
        de = torch.tanh(description_embeddings @ P["description_embeddings_linear.0.weight"].T
                        + P["description_embeddings_linear.0.bias"])
        encoded = torch.cat([encoded, de[:, None, :].expand(-1, encoded.shape[1], -1)], 2)
    att_encoded = encoded @ P["att_encoder.weight"].T
    return encoded, att_encoded


def prenet_fwd(P, x: Tensor, drop1: Optional[Tensor], drop2: Optional[Tensor]) -> Tensor:
    """model/tacotron2.py:85-92; dropout always active (model/modules.py:10-12)."""
    y = torch.relu(x @ P["prenet.0.weight"].T)
    if drop1 is not None:
        y = y * drop1
    y = torch.relu(y @ P["prenet.3.weight"].T)
    if drop2 is not None:
        y = y * drop2
    return y


def attention_fwd(P, att_h: Tensor, memory: Tensor, processed_memory: Tensor, w_cat: Tensor, mask: Tensor):
    """model/attention.py:52-69.  w_cat (B,2,L); mask (B,L) bool True=padding."""
    q = att_h @ P["decoder.attention.query_layer.weight"].T                  # (B,Ad)
    Wc = P["decoder.attention.location_conv.weight"]                         # (F,2,K)
    K = Wc.shape[2]
    pad = (K - 1) // 2
    B, _, L = w_cat.shape
    wp = torch.zeros(B, 2, L + 2 * pad, dtype=w_cat.dtype)
    wp[:, :, pad:pad + L] = w_cat
    win = wp.unfold(2, K, 1)                                                 # (B,2,L,K)
    conv = torch.einsum("bclk,fck->blf", win, Wc)                            # (B,L,F)
    loc = conv @ P["decoder.attention.location_dense.weight"].T              # (B,L,Ad)
    e = torch.tanh(q[:, None, :] + loc + processed_memory) @ P["decoder.attention.v.weight"][0]
    e = e.masked_fill(mask, float("-inf"))
    m = e.max(1, keepdim=True).values
    p = torch.exp(e - m)
    w = p / p.sum(1, keepdim=True)
    ctx = torch.einsum("bl,ble->be", w, memory)
    return ctx, w


def decoder_step(P, prenet_out, att_h, att_c, ctx, w, w_cum, dec_h, dec_c, memory, processed_memory, mask,
                 att_drop: Optional[Tensor], dec_drop: Optional[Tensor], extra_decoder_in: Optional[Tensor] = None):
    """model/decoder.py:68-119.  Returns new states; the dropped h is the carried h (Appendix C.1).
    extra_decoder_in (the controls vector, model/tacotron2.py:279-286) is appended to the decoder-LSTM input and to the
    mel projection input, not to the gate projection (model/decoder.py:94-109)."""
    g = torch.cat([prenet_out, ctx], -1) @ P["decoder.att_rnn.weight_ih"].T + P["decoder.att_rnn.bias_ih"] \
        + att_h @ P["decoder.att_rnn.weight_hh"].T + P["decoder.att_rnn.bias_hh"]
    att_h, att_c = lstm_cell(g, att_c)
    if att_drop is not None:
        att_h = att_h * att_drop
    ctx, w = attention_fwd(P, att_h, memory, processed_memory, torch.stack([w, w_cum], 1), mask)
    w_cum = w_cum + w
    xe = [extra_decoder_in] if extra_decoder_in is not None else []
    g = torch.cat([att_h, ctx] + xe, -1) @ P["decoder.lstm.weight_ih"].T + P["decoder.lstm.bias_ih"] \
        + dec_h @ P["decoder.lstm.weight_hh"].T + P["decoder.lstm.bias_hh"]
    dec_h, dec_c = lstm_cell(g, dec_c)
    if dec_drop is not None:
        dec_h = dec_h * dec_drop
    hc = torch.cat([dec_h, ctx], -1)
    gate = hc @ P["decoder.gate.weight"].T + P["decoder.gate.bias"]
    mel = torch.cat([hc] + xe, -1) @ P["decoder.mel_out.weight"].T + P["decoder.mel_out.bias"]
    return mel, gate, att_h, att_c, ctx, w, w_cum, dec_h, dec_c


def postnet_fwd(P, mels: Tensor, training: bool, post_drop: Optional[List[Tensor]] = None,
                new_stats: Optional[dict] = None) -> Tensor:
    """model/postnet.py:5-52 on channel-last (B,T,M)."""
    x = mels
    for li in range(5):
        x = conv1d_cl(x, P[f"postnet.postnet.{4 * li}.weight"], None)
        k = f"postnet.postnet.{4 * li + 1}"
        x = batch_norm_cl(x, P[k + ".weight"], P[k + ".bias"], P[k + ".running_mean"], P[k + ".running_var"],
                          training, new_stats, k)
        if li < 4:
            x = torch.tanh(x)
        if post_drop is not None and post_drop[li] is not None:
            x = x * post_drop[li]
    return x


# --------------------------------------------------------------------------------------------
# full forward
# --------------------------------------------------------------------------------------------
def tacotron2_fwd(P, d: dict, chars_idx: Tensor, chars_len: Tensor, teacher_forcing: bool,
                  mel: Optional[Tensor] = None, mel_len: Optional[Tensor] = None, speaker_id=None,
                  description_embeddings=None, max_len_override: Optional[int] = None,
                  training: bool = True, masks: Optional[dict] = None, new_stats: Optional[dict] = None,
                  trace: Optional[dict] = None, controls: Optional[Tensor] = None):
    """model/tacotron2.py:155-347.  ``masks`` keys (all optional):
       enc_drop [3x(B,L,E)], prenet_drop (TF: [2x(B,T+1,P)]; inference: list per step of [2x(B,P)],
       entry 0 is for the initial zero frame), att_drop (T,B,A), dec_drop (T,B,D), post_drop [5x(B,T,C)].
    Returns (mels, mels_post, gates, alignments) as the reference does."""
    masks = masks or {}
    assert bool(d.get("controls")) == (controls is not None), "controls tensor and the controls flag must agree (model/tacotron2.py:185-191)"
    dt = P["prenet.0.weight"].dtype
    B, L = chars_idx.shape
    encoded = encoder_fwd(P, chars_idx, chars_len, training, masks.get("enc_drop"), new_stats)
    memory, pm = condition(P, d, encoded, speaker_id, description_embeddings)
    mask = torch.arange(L)[None, :] >= chars_len[:, None]
    A, D, Ef = d["att_rnn_dim"], d["rnn_hidden_dim"], memory.shape[2]
    att_h = torch.zeros(B, A, dtype=dt); att_c = torch.zeros(B, A, dtype=dt)
    ctx = torch.zeros(B, Ef, dtype=dt)
    w = torch.zeros(B, memory.shape[1], dtype=dt); w_cum = torch.zeros_like(w)
    dec_h = torch.zeros(B, D, dtype=dt); dec_c = torch.zeros(B, D, dtype=dt)
    if max_len_override is not None:
        max_len = max_len_override
    elif mel is not None:
        max_len = mel.shape[1]
    else:
        raise Exception("If Mel spectrogram is not given, max_len_override is required!")
    pd = masks.get("prenet_drop")
    if teacher_forcing:
        dec_in = torch.cat([torch.zeros(B, 1, mel.shape[2], dtype=dt), mel], 1)
        dec_in = prenet_fwd(P, dec_in, pd[0] if pd else None, pd[1] if pd else None)
        prev = dec_in[:, 0]
        lengths = mel_len.to(torch.int64)
    else:
        prev = prenet_fwd(P, torch.zeros(B, d["num_mels"], dtype=dt),
                          pd[0][0] if pd else None, pd[0][1] if pd else None)
        done = torch.zeros(B, dtype=torch.bool)
        lengths = torch.zeros(B, dtype=torch.int64)
    mels, gates, aligns = [], [], []
    ad, dd = masks.get("att_drop"), masks.get("dec_drop")
    for i in range(max_len):
        mel_o, gate_o, att_h, att_c, ctx, w, w_cum, dec_h, dec_c = decoder_step(
            P, prev, att_h, att_c, ctx, w, w_cum, dec_h, dec_c, memory, pm, mask,
            ad[i] if ad is not None else None, dd[i] if dd is not None else None, extra_decoder_in=controls)
        mels.append(mel_o); gates.append(gate_o); aligns.append(w)
        if trace is not None:
            trace.setdefault("att_h", []).append(att_h); trace.setdefault("ctx", []).append(ctx)
            trace.setdefault("dec_h", []).append(dec_h)
        if teacher_forcing:
            prev = dec_in[:, i + 1]
        else:
            g = gate_o[:, 0]
            done = done | (g < 0.0)
            lengths = lengths + (g >= 0.0).to(torch.int64)
            if bool(done.all()):
                break
            prev = prenet_fwd(P, mel_o.detach(), pd[i + 1][0] if pd else None, pd[i + 1][1] if pd else None)
    mels = torch.stack(mels, 1); gates = torch.stack(gates, 1); aligns = torch.stack(aligns, 1)
    if trace is not None:
        trace["memory"], trace["pm"], trace["encoded"], trace["lengths"] = memory, pm, encoded, lengths
    post = mels + postnet_fwd(P, mels, training, masks.get("post_drop"), new_stats)
    mm = (torch.arange(mels.shape[1])[None, :] >= lengths[:, None])[:, :, None]
    mels = mels.masked_fill(mm, 0.0)
    post = post.masked_fill(mm, 0.0)
    gates = gates.masked_fill(mm, -1000.0)
    return mels, post, gates, aligns


def tts_loss(mels, post, gates, mel_tgt, gate_tgt):
    """model/tts_model.py:197-201: plain means over all elements, padding included."""
    x, y = gates, gate_tgt
    bce = (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()
    mel_l = ((mels - mel_tgt) ** 2).mean()
    post_l = ((post - mel_tgt) ** 2).mean()
    return bce + mel_l + post_l, bce, mel_l, post_l


def adam_l2_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
                 b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam with weight_decay = L2-in-gradient (model/tts_model.py:78-81)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    p = p - lr * mhat / (vhat.sqrt() + eps)
    return p, m, v


def clip_coef(grads: List[Tensor], max_norm: float = 1.0) -> float:
    """Lightning gradient_clip_val=1.0 -> torch clip_grad_norm_ (run/train.py:240)."""
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (tot + 1e-6)), tot
