"""HiFi-GAN generator forward (the vocoder `say --hifi-gan-checkpoint` uses, run/say.py:66-86,153-159) on the gfx950 GEMM
kernels: a drop-in for the reference's model.hifi_gan.Generator at inference (model/hifi_gan.py:154-216), same state_dict
keys (weight-normed `weight_g` / `weight_v` as in the published UNIVERSAL_V1 checkpoint, or plain `weight`).

Everything is channel-last (T, C) with zero margins of PAD rows on both sides, one utterance at a time:
  * Conv1d(k, dilation d, 'same'): d = 1 is ONE GEMM over overlapping rows (lda = C < K = k*C, as the Tacotron convolutions);
    d > 1 is ONE GEMM too - the K axis of an A row is k blocks of C channels that lie d rows apart (T2Gemm.a_tap_len / a_tap_stride) -
    for channel counts that are multiples of 32 (every layer of the published configurations), else k GEMMs over rows shifted by
    j*d that accumulate into the output (B = the packed weight's tap-j column block);
  * ConvTranspose1d(k = 2u, stride u, padding u/2) - every upsampling layer of HiFi-GAN - is ONE GEMM: output position
    o = q*u + r - u/2 receives x[q] . W[:, :, r] + x[q-1] . W[:, :, r+u], so the A rows are the overlapping pairs
    [x[q-1] | x[q]] (K = 2*Cin) and the N = u*Cout output columns of row q are u consecutive output positions;
  * leaky-ReLU / the 1/num_kernels average / residual sums: t2_leaky_relu, t2_axpy, the GEMM's accumulate epilogue; tanh:
    t2_tanh_bias.
Parity: tests/golden/hifigan.npz (generated from the reference's Generator by oracle/make_golden_hifigan.py)."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import torch

from ._lib import call
from .engine import _ptr, _stream, gemm

PAD = 32          # zero margin rows: >= the farthest tap of any layer (k = 11, d = 5 reaches 25 rows) and the upsampler's u/2
LRELU_SLOPE = 0.1

UNIVERSAL_V1 = dict(resblock="1", upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], upsample_initial_channel=512,
                    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """torch.nn.utils.remove_weight_norm on a state_dict: w = g * v / ||v|| (norm over all dims but 0)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_v"):
            g = sd[k[:-1] + "g"].float()
            v = v.float()
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            out[k[:-2]] = g * v / norm
        elif not k.endswith("weight_g"):
            out[k] = v.float()
    return out


class _Conv:
    """Conv1d weights packed for the GEMM: wp[co][j*Ci + ci] = w[co][ci][j]."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor, dilation: int, dev):
        self.Co, self.Ci, self.k = w.shape
        self.d = dilation
        self.p = (self.k * dilation - dilation) // 2
        assert self.p <= PAD
        self.wp = w.permute(0, 2, 1).reshape(self.Co, self.k * self.Ci).contiguous().to(dev)
        self.b = b.contiguous().to(dev)

    def __call__(self, x: torch.Tensor, y: torch.Tensor, L: int, accumulate: bool = False):
        """x, y: (L + 2*PAD, C) margin layouts.  y[PAD + t] (+)= bias + sum_j x[PAD + t + j*d - p] . w_j."""
        Ci, Co, k = self.Ci, self.Co, self.k
        yo = _ptr(y, PAD * Co)
        if self.d == 1:
            gemm(_ptr(x, (PAD - self.p) * Ci), self.wp, yo, L, Co, k * Ci, Ci, k * Ci, Co, bias=self.b, accumulate=1 if accumulate else 0)
            return
        if Ci % 32 == 0:     # dilated: still ONE GEMM - the A row's K axis is k blocks of Ci channels, d rows apart (T2Gemm.a_tap_len)
            gemm(_ptr(x, (PAD - self.p) * Ci), self.wp, yo, L, Co, k * Ci, Ci, k * Ci, Co, bias=self.b,
                 accumulate=1 if accumulate else 0, a_tap_len=Ci, a_tap_stride=self.d * Ci)
            return
        for j in range(k):
            gemm(_ptr(x, (PAD + j * self.d - self.p) * Ci), _ptr(self.wp, j * Ci), yo, L, Co, Ci, Ci, k * Ci, Co,
                 bias=self.b if j == 0 else None, accumulate=1 if (accumulate or j > 0) else 0)


class _Up:
    """ConvTranspose1d(k = 2u, stride u, padding u/2) packed as wt[r*Co + co][tap_block*Ci + ci]."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor, u: int, dev):
        self.Ci, self.Co, k = w.shape
        if k != 2 * u or u % 2:
            raise NotImplementedError(f"ConvTranspose1d with kernel {k}, stride {u}: only k = 2*stride with even stride (every "
                                      "published HiFi-GAN configuration) is laid out as one GEMM")
        self.u = u
        wt = torch.empty(u, self.Co, 2, self.Ci)
        wt[:, :, 0, :] = w[:, :, u:].permute(2, 1, 0)          # x[q-1] meets tap r + u
        wt[:, :, 1, :] = w[:, :, :u].permute(2, 1, 0)          # x[q]   meets tap r
        self.wt = wt.reshape(u * self.Co, 2 * self.Ci).contiguous().to(dev)
        self.b = b.repeat(u).contiguous().to(dev)

    def __call__(self, x: torch.Tensor, y: torch.Tensor, L: int):
        """x (L + 2*PAD, Ci) -> y (L*u + 2*PAD, Co); y's margins are re-zeroed (rows q = 0 and q = L spill u/2 rows into them)."""
        Ci, Co, u = self.Ci, self.Co, self.u
        p = u // 2
        gemm(_ptr(x, (PAD - 1) * Ci), self.wt, _ptr(y, (PAD - p) * Co), L + 1, u * Co, 2 * Ci, Ci, 2 * Ci, u * Co, bias=self.b)
        y[:PAD].zero_(); y[PAD + L * u:].zero_()


class Generator:
    """forward(mel (B, 80, T) or (80, T)) -> waveform (B, 1, T * prod(upsample_rates)), as the reference's Generator in eval mode
    after remove_weight_norm()."""

    def __init__(self, h: Optional[dict] = None, device="cuda:0"):
        self.h = dict(UNIVERSAL_V1 if h is None else h)
        self.dev = torch.device(device)
        self.rates: List[int] = list(self.h["upsample_rates"])
        self.nk = len(self.h["resblock_kernel_sizes"])
        self.res1 = str(self.h["resblock"]) == "1"
        self.loaded = False

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        sd = fold_weight_norm({k: torch.as_tensor(v) for k, v in sd.items()})
        dev, h = self.dev, self.h
        self.conv_pre = _Conv(sd["conv_pre.weight"], sd["conv_pre.bias"], 1, dev)
        self.ups = [_Up(sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], u, dev) for i, u in enumerate(self.rates)]
        self.blocks = []
        for i in range(len(self.rates)):
            for j, dil in enumerate(h["resblock_dilation_sizes"]):
                pre = f"resblocks.{i * self.nk + j}."
                if self.res1:
                    pairs = [(_Conv(sd[pre + f"convs1.{c}.weight"], sd[pre + f"convs1.{c}.bias"], d, dev),
                              _Conv(sd[pre + f"convs2.{c}.weight"], sd[pre + f"convs2.{c}.bias"], 1, dev)) for c, d in enumerate(dil)]
                else:
                    pairs = [(_Conv(sd[pre + f"convs.{c}.weight"], sd[pre + f"convs.{c}.bias"], d, dev), None) for c, d in enumerate(dil)]
                self.blocks.append(pairs)
        self.conv_post = _Conv(sd["conv_post.weight"], sd["conv_post.bias"], 1, dev)
        self.loaded = True
        return self

    @classmethod
    def from_checkpoint(cls, path: str, device="cuda:0"):
        """run/say.py:76-86: `config.json` next to the checkpoint file (UNIVERSAL_V1 values when absent), weights under "generator"."""
        cfg = os.path.join(os.path.dirname(os.path.abspath(path)), "config.json")
        h = json.load(open(cfg)) if os.path.exists(cfg) else None
        ck = torch.load(path, map_location="cpu", weights_only=True)
        return cls(h, device).load_state_dict(ck["generator"] if "generator" in ck else ck)

    def _buf(self, L: int, C: int) -> torch.Tensor:
        return torch.zeros(L + 2 * PAD, C, dtype=torch.float32, device=self.dev)

    def _lrelu(self, x, y, scale=1.0, slope=LRELU_SLOPE):
        call("t2_leaky_relu", x, y, x.numel(), float(scale), float(slope), _stream())

    def _one(self, mel: torch.Tensor) -> torch.Tensor:
        """mel (80, T) -> (T * prod(rates),)"""
        L = mel.shape[1]
        x0 = self._buf(L, mel.shape[0])
        x0[PAD:PAD + L].copy_(mel.t())
        x = self._buf(L, self.conv_pre.Co)
        self.conv_pre(x0, x, L)
        for i, up in enumerate(self.ups):
            a = torch.empty_like(x)
            self._lrelu(x, a)                     # margins: lrelu(0) = 0
            L2 = L * up.u
            x = self._buf(L2, up.Co)
            up(a, x, L)
            L = L2
            xs = self._buf(L, up.Co)
            t1, t2 = torch.empty_like(x), self._buf(L, up.Co)
            for j in range(self.nk):
                r = x.clone()
                for c1, c2 in self.blocks[i * self.nk + j]:
                    self._lrelu(r, t1)
                    if c2 is None:
                        c1(t1, r, L, accumulate=True)            # x = c(lrelu(x)) + x
                    else:
                        c1(t1, t2, L)
                        self._lrelu(t2, t1)
                        c2(t1, r, L, accumulate=True)            # x = c2(lrelu(c1(lrelu(x)))) + x
                call("t2_axpy", r, xs, r.numel(), 1.0, _stream())
            x = xs
            if i + 1 < len(self.ups):
                a = torch.empty_like(x)
                self._lrelu(x, a, scale=1.0 / self.nk, slope=1.0)      # x = xs / num_kernels (slope 1: a plain scale)
                x = a
        a = torch.empty_like(x)
        self._lrelu(x, a, scale=1.0 / self.nk, slope=0.01)             # xs / num_kernels, then F.leaky_relu's default slope
        y = self._buf(L, 1)
        self.conv_post(a, y, L)
        wav = y[PAD:PAD + L, 0].contiguous()
        call("t2_tanh_bias", wav, None, L, 1, _stream())
        return wav

    def __call__(self, mel: torch.Tensor) -> torch.Tensor:
        assert self.loaded, "load_state_dict first"
        if not mel.is_cuda:
            raise RuntimeError("hifigan.Generator needs CUDA/HIP tensors: the product path has no CPU fallback")
        if mel.dim() == 2:
            mel = mel[None]
        with torch.no_grad():
            return torch.stack([self._one(m.float().contiguous()) for m in mel])[:, None, :]

    forward = __call__
