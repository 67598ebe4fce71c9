#!/usr/bin/env python3
"""Generate tests/golden/hifigan.npz by running the REFERENCE's HiFi-GAN Generator (imported from /root/reference, needs
only torch) on CPU.  Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_hifigan.py

Two small generators with seeded random weights (the reference's N(0, 0.01) initialisation gives outputs near zero, so the
weight-norm factors are redrawn at a visible scale): fixture entries are tensors only - the weight-normed state_dict
(`*.weight_g`, `*.weight_v`, `*.bias`: the layout of the published UNIVERSAL_V1 checkpoint run/say.py:76-82 loads), the
input log-mel (1, 80, T) and the reference's output waveform after remove_weight_norm() + eval() (run/say.py:84-86,153-159).
  v1: resblock "1" (ResBlock1, model/hifi_gan.py:20-104), two upsampling stages, kernels 3 and 5
  v2: resblock "2" (ResBlock2, :107-151), stride-8 first stage (k=16: the UNIVERSAL_V1 shape of the transposed convolution)
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")

from model.hifi_gan import Generator  # noqa: E402  (the reference)


class AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


CONFIGS = {
    "v1": dict(resblock="1", upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=32,
               resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]]),
    "v2": dict(resblock="2", upsample_rates=[8, 2], upsample_kernel_sizes=[16, 4], upsample_initial_channel=16,
               resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3], [1, 2]]),
}


def main():
    out = {}
    for name, cfg in CONFIGS.items():
        torch.manual_seed(7 if name == "v1" else 11)
        g = Generator(AttrDict(cfg))
        gen = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for k, p in g.named_parameters():
                if k.endswith("weight_v"):
                    p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
                elif k.endswith("weight_g"):
                    p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
                elif k.endswith("bias"):
                    p.copy_(torch.randn(p.shape, generator=gen) * 0.1)
        sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
        T = 13 if name == "v1" else 9
        mel = torch.randn(1, 80, T, generator=gen) * 1.5 - 4.0
        g.remove_weight_norm()
        g.eval()
        with torch.no_grad():
            wav = g(mel)
        for k, v in sd.items():
            out[f"{name}.sd.{k}"] = v.numpy()
        out[f"{name}.mel"] = mel.numpy()
        out[f"{name}.wav"] = wav.numpy()
        for k, v in cfg.items():
            out[f"{name}.cfg.{k}"] = np.array(v if not isinstance(v, str) else int(v))
        print(name, "keys", len(sd), "wav", tuple(wav.shape), "abs max", float(wav.abs().max()), "std", float(wav.std()))
    np.savez_compressed(os.path.join(OUT, "hifigan.npz"), **out)
    print("wrote", os.path.join(OUT, "hifigan.npz"))


if __name__ == "__main__":
    main()
