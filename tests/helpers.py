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
