"""Shared test helpers: golden-fixture loading and mask plumbing (CPU only, no GPU needed)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SMALL = dict(num_chars=39, encoded_dim=32, encoder_kernel_size=5, num_mels=16, prenet_dim=16,
             att_rnn_dim=32, att_dim=16, rnn_hidden_dim=32, postnet_dim=32)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def params_from(z, prefix="p.", dtype=torch.float32):
    P = {}
    for k, v in z.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v))
            P[k[len(prefix):]] = t.to(dtype) if t.is_floating_point() else t
    return P


def tf_masks_from(z, dtype=torch.float32):
    t = lambda a: torch.from_numpy(a).to(dtype)
    return dict(
        enc_drop=[t(z[f"m.enc_drop.{i}"]) for i in range(3)],
        prenet_drop=[t(z[f"m.prenet_drop.{i}"]) for i in range(2)],
        att_drop=t(z["m.att_drop"]), dec_drop=t(z["m.dec_drop"]),
        post_drop=[t(z[f"m.post_drop.{i}"]) for i in range(5)],
    )


def write_hifigan_checkpoint(hdir, n_mels=80, ch0=16, seed=0):
    """A generator checkpoint in the published layout ({"generator": weight-normed state_dict}, config.json next to it):
    UNIVERSAL_V1 strides 8*8*2*2 = 256 samples per frame, narrow channels for speed.  Returns the checkpoint path."""
    import json
    os.makedirs(hdir, exist_ok=True)
    hcfg = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=ch0,
                resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    with open(os.path.join(hdir, "config.json"), "w") as f:
        json.dump(hcfg, f)
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def wn(name, *shape):
        sd[name + ".weight_v"] = torch.randn(*shape, generator=g) * 0.2
        sd[name + ".weight_g"] = torch.rand(shape[0], 1, 1, generator=g) * 0.5 + 0.25
        sd[name + ".bias"] = torch.zeros(shape[0] if "ups" not in name else shape[1])
    wn("conv_pre", ch0, n_mels, 7)
    ch = ch0
    for i, k in enumerate(hcfg["upsample_kernel_sizes"]):
        wn(f"ups.{i}", ch, ch // 2, k); ch //= 2
        for j, kk in enumerate(hcfg["resblock_kernel_sizes"]):
            for c in range(3):
                wn(f"resblocks.{i * 3 + j}.convs1.{c}", ch, ch, kk); wn(f"resblocks.{i * 3 + j}.convs2.{c}", ch, ch, kk)
    wn("conv_post", 1, ch, 7)
    path = os.path.join(hdir, "g_00000001")
    torch.save({"generator": sd}, path)
    return path


def dekink_masks(P, d, chars_idx, mel, masks, eps=1e-5, training=True):
    """ReLU has a jump in its derivative at 0: an element whose pre-activation is within fp32 rounding of zero can take
    either derivative depending on the summation order of the kernel that produced it, and the whole upstream gradient
    then differs legitimately between two correct fp32 implementations (seen once: tools/debug_b35.py).  This takes such
    elements out of BOTH sides of a gradient comparison: the dropout scale mask that multiplies the ReLU output is set to
    zero wherever the oracle's pre-activation is within `eps` of zero (encoder BN-ReLU layers, model/encoder.py:33-44;
    prenet ReLUs, model/tacotron2.py:85-92).  Layers are handled in order, so a changed mask is seen by the next layer's
    pre-activations.  Returns (masks with the kink elements dropped, number of elements dropped)."""
    from oracle import tacotron2_ref as R
    out = dict(masks)
    dropped = 0
    with torch.no_grad():
        ed = masks.get("enc_drop")
        if ed is not None:
            ed = [m.clone() for m in ed]
            x = P["encoder.embedding.weight"][chars_idx]
            for li, i in enumerate((0, 4, 8)):
                x = R.conv1d_cl(x, P[f"encoder.convolutions.{i}.weight"], P[f"encoder.convolutions.{i}.bias"])
                k = f"encoder.convolutions.{i + 1}"
                x = R.batch_norm_cl(x, P[k + ".weight"], P[k + ".bias"], P[k + ".running_mean"], P[k + ".running_var"],
                                    training, None, k)
                kink = x.abs() < eps
                dropped += int((kink & (ed[li] != 0)).sum())
                ed[li][kink] = 0.0
                x = torch.relu(x) * ed[li]
            out["enc_drop"] = ed
        pd = masks.get("prenet_drop")
        if pd is not None and mel is not None:
            pd = [m.clone() for m in pd]
            B, T, M = mel.shape
            x = torch.cat([torch.zeros(B, 1, M, dtype=mel.dtype), mel], 1)
            for li, name in enumerate(("prenet.0.weight", "prenet.3.weight")):
                x = x @ P[name].T
                kink = (x.abs() < eps) & (x != 0)        # exact zeros (the initial zero frame) have no upstream gradient
                dropped += int((kink & (pd[li] != 0)).sum())
                pd[li][kink] = 0.0
                x = torch.relu(x) * pd[li]
            out["prenet_drop"] = pd
    return out, dropped
