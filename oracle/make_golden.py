#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported from /root/reference) on CPU.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
The reference never travels to the GPU box; only these small data fixtures do.  Nothing from the
reference's source text is stored - only tensors (weights drawn by its own initialisers, inputs,
the dropout masks it drew, and its outputs / gradients).

Dropout masks are captured by wrapping ``torch.nn.functional.dropout`` while the reference runs
(nn.Dropout.forward -> F.dropout), so the oracle and the HIP path can replay the same draws.
Call order: encoder conv 0,1,2; prenet 1,2; per frame att_rnn_dropout, lstm_dropout
(inference: then prenet 1,2 for the next frame); postnet 0..4.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")

from model.tacotron2 import Tacotron2  # noqa: E402  (the reference)

SMALL = dict(num_chars=39, encoded_dim=32, encoder_kernel_size=5, num_mels=16, prenet_dim=16,
             att_rnn_dim=32, att_dim=16, rnn_hidden_dim=32, postnet_dim=32)

_captured = []
_orig_dropout = F.dropout


def _recording_dropout(input, p=0.5, training=True, inplace=False):
    if not training or p == 0.0:
        return input
    keep = torch.bernoulli(torch.full_like(input, 1.0 - p))
    scale = keep / (1.0 - p)
    _captured.append(scale.clone())
    return input * scale


def build(dropout, seed, **extra):
    torch.manual_seed(seed)
    m = Tacotron2(dropout=dropout, **SMALL, **extra)
    # non-trivial BN running stats / affine so eval mode exercises them
    g = torch.Generator().manual_seed(seed + 1)
    for name, buf in m.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(torch.rand(buf.shape, generator=g) * 0.2 - 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.75)
    for name, p in m.named_parameters():
        if (".convolutions." in name or "postnet.postnet." in name) and p.dim() == 1:
            idx = int(name.split(".")[2])
            if idx % 4 == 1:
                with torch.no_grad():
                    if name.endswith("weight"):
                        p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
                    else:
                        p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)
    return m


def make_batch(seed, lens, tlens, num_mels, chars=39):
    g = torch.Generator().manual_seed(seed)
    B, L, T = len(lens), max(lens), max(tlens)
    ci = torch.zeros(B, L, dtype=torch.int64)
    mel = torch.zeros(B, T, num_mels)
    gate = torch.zeros(B, T, 1)
    for b in range(B):
        ci[b, :lens[b]] = torch.randint(1, chars + 1, (lens[b],), generator=g)
        mel[b, :tlens[b]] = torch.randn(tlens[b], num_mels, generator=g) * 1.5 - 3.0
        gate[b, :tlens[b]] = 1.0
        gate[b, tlens[b] - 1] = 0.0
    return ci, torch.tensor(lens, dtype=torch.int64), mel, torch.tensor(tlens, dtype=torch.int32), gate


def sd_np(m, prefix="p."):
    return {prefix + k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def masks_tf(cap, T):
    """captured list -> oracle mask dict (channel-last, time-major step masks)."""
    i = 0
    out = {}
    for li in range(3):
        out[f"m.enc_drop.{li}"] = cap[i].transpose(1, 2).contiguous().numpy(); i += 1
    for li in range(2):
        out[f"m.prenet_drop.{li}"] = cap[i].numpy(); i += 1
    att, dec = [], []
    for _ in range(T):
        att.append(cap[i]); dec.append(cap[i + 1]); i += 2
    out["m.att_drop"] = torch.stack(att, 0).numpy()
    out["m.dec_drop"] = torch.stack(dec, 0).numpy()
    for li in range(5):
        out[f"m.post_drop.{li}"] = cap[i].transpose(1, 2).contiguous().numpy(); i += 1
    assert i == len(cap), (i, len(cap))
    return out


def case_tf_eval():
    m = build(0.0, 11).eval()
    ci, cl, mel, ml, gate = make_batch(21, [17, 11, 5], [23, 15, 9], SMALL["num_mels"])
    with torch.no_grad():
        mels, post, gates, al = m(ci, cl, True, mel, ml)
        enc = m.encoder(ci, cl)
    d = sd_np(m)
    d.update(chars_idx=ci.numpy(), chars_len=cl.numpy(), mel=mel.numpy(), mel_len=ml.numpy(), gate=gate.numpy(),
             o_mels=mels.numpy(), o_post=post.numpy(), o_gates=gates.numpy(), o_align=al.numpy(),
             o_encoded=enc.numpy())
    np.savez_compressed(os.path.join(OUT, "tf_eval.npz"), **d)


def case_tf_train(name, seed, extra, lens, tlens):
    m = build(0.5, seed, **extra).train()
    ci, cl, mel, ml, gate = make_batch(seed + 10, lens, tlens, SMALL["num_mels"])
    kw = {}
    g = torch.Generator().manual_seed(seed + 20)
    if extra.get("speaker_tokens"):
        kw["speaker_id"] = torch.randint(0, extra["num_speakers"], (len(lens),), generator=g, dtype=torch.int32)
    if extra.get("description_embeddings"):
        kw["description_embeddings"] = torch.randn(len(lens), extra["description_embeddings_dim"], generator=g)
    if extra.get("controls"):
        kw["controls"] = torch.randn(len(lens), extra["controls_dim"], generator=g)
    _captured.clear()
    F.dropout = _recording_dropout
    torch.nn.functional.dropout = _recording_dropout
    try:
        mels, post, gates, al = m(ci, cl, True, mel, ml, **kw)
    finally:
        F.dropout = _orig_dropout
        torch.nn.functional.dropout = _orig_dropout
    # loss exactly as model/tts_model.py:197-201
    gate_loss = F.binary_cross_entropy_with_logits(gates, gate)
    mel_loss = F.mse_loss(mels, mel)
    post_loss = F.mse_loss(post, mel)
    loss = gate_loss + mel_loss + post_loss
    loss.backward()
    d = {}
    # parameters BEFORE the step are what produced the outputs; BN buffers were updated in place by the
    # forward, so store the pre-forward values separately (rebuild the same model for them)
    m0 = build(0.5, seed, **extra)
    d.update(sd_np(m0))
    d.update({"new." + k: v.detach().numpy() for k, v in m.state_dict().items()
              if k.endswith(("running_mean", "running_var", "num_batches_tracked"))})
    d.update({"g." + k: p.grad.numpy() for k, p in m.named_parameters()})
    d.update(masks_tf(list(_captured), max(tlens)))
    d.update(chars_idx=ci.numpy(), chars_len=cl.numpy(), mel=mel.numpy(), mel_len=ml.numpy(), gate=gate.numpy(),
             o_mels=mels.detach().numpy(), o_post=post.detach().numpy(), o_gates=gates.detach().numpy(),
             o_align=al.detach().numpy(),
             o_loss=np.array([loss.item(), gate_loss.item(), mel_loss.item(), post_loss.item()], np.float64))
    for k, v in kw.items():
        d[k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def case_infer(name="infer", seed=31, extra=None):
    extra = extra or {}
    m = build(0.5, seed, **extra).eval()
    kw = {}
    if extra.get("controls"):
        kw["controls"] = torch.randn(3, extra["controls_dim"], generator=torch.Generator().manual_seed(seed + 5))
    # nudge the stop projection so samples stop at different frames (random weights never stop)
    with torch.no_grad():
        m.decoder.gate.bias.fill_(0.05)
        m.decoder.gate.weight.mul_(3.0)
    ci, cl, _, _, _ = make_batch(41, [9, 14, 6], [5, 5, 5], SMALL["num_mels"])
    _captured.clear()
    F.dropout = _recording_dropout
    torch.nn.functional.dropout = _recording_dropout
    try:
        with torch.no_grad():
            mels, post, gates, al = m(ci, cl, False, max_len_override=40, **kw)
    finally:
        F.dropout = _orig_dropout
        torch.nn.functional.dropout = _orig_dropout
    cap = list(_captured)
    # eval mode: only the prenet draws (2 per prenet call): call 0 for the zero frame, then one per fed-back frame
    assert len(cap) % 2 == 0
    pm = torch.stack([torch.stack([cap[2 * i], cap[2 * i + 1]], 0) for i in range(len(cap) // 2)], 0)
    d = sd_np(m)
    d.update(chars_idx=ci.numpy(), chars_len=cl.numpy(), **{"m.prenet_drop": pm.numpy()},
             o_mels=mels.numpy(), o_post=post.numpy(), o_gates=gates.numpy(), o_align=al.numpy(),
             max_len=np.array(40))
    for k, v in kw.items():
        d[k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    case_tf_eval()
    case_tf_train("tf_train", 51, {}, [17, 11, 5], [23, 15, 9])
    case_tf_train("tf_train_desc", 61, dict(speaker_tokens=True, num_speakers=7, description_embeddings=True,
                                            description_embeddings_dim=24), [12, 16], [13, 19])
    case_infer()
    # prosody-controls extension (model/decoder.py:40-48,97-109; model/tacotron2.py:279-286)
    case_tf_train("tf_train_ctrl", 71, dict(controls=True, controls_dim=5), [14, 9, 6], [17, 11, 8])
    case_infer("infer_ctrl", 86, dict(controls=True, controls_dim=5))   # stops per sample at different frames; runs to the cap
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
