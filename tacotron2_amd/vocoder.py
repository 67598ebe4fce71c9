"""Griffin-Lim vocoding for `say` (run/say.py:161-171: when no HiFi-GAN checkpoint is given the
reference calls librosa.feature.inverse.mel_to_audio(exp(mel).T, n_fft=1024, hop_length=256, win_length=1024, center=True,
power=1.0, fmin=0, fmax=8000), i.e. mel -> linear magnitude -> 32 Griffin-Lim iterations with momentum 0.99).

Both transforms of the iteration are dense contractions with a fixed DFT basis, so they run on the MFMA GEMM of the C ABI
(t2_gemm): the analysis reads the padded signal as overlapping rows (lda = hop), the synthesis multiplies by the
window-folded inverse basis and overlap-adds.  Elementwise phase updates are torch glue.

PARITY UNPINNED: librosa is not importable here and the reference holds no audio fixture; Griffin-Lim starts from random
phases, so only properties are tested (exact STFT -> ISTFT round trip, spectral convergence).  The mel -> linear step uses the
clamped minimum-norm solution (pseudo-inverse of the slaney filterbank) where librosa solves a non-negative least squares
problem per frame.
"""
from __future__ import annotations

import wave

import numpy as np
import torch

from .datasets.logmel import TacotronMelSpectrogram
from .engine import gemm


class GriffinLim:
    def __init__(self, n_mels: int = 80, sample_rate: int = 22050, n_fft: int = 1024, hop_length: int = 256,
                 f_min: float = 0.0, f_max: float = 8000.0, n_iter: int = 32, momentum: float = 0.99, device="cuda:0"):
        self.front = TacotronMelSpectrogram(n_mels, sample_rate, n_fft, hop_length, f_min, f_max, device)
        self.sr, self.n_fft, self.hop, self.n_iter, self.momentum = sample_rate, n_fft, hop_length, n_iter, momentum
        self.device = self.front.device
        nb = self.nb = n_fft // 2 + 1
        n = np.arange(n_fft)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
        ang = 2 * np.pi * np.outer(np.arange(nb), n) / n_fft
        herm = np.full(nb, 2.0)
        herm[0] = herm[-1] = 1.0
        # x[n] = (1/N) sum_k herm_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)), times the synthesis window
        ib = np.concatenate([np.cos(ang) * herm[:, None], -np.sin(ang) * herm[:, None]], 0) * (win[None, :] / n_fft)
        self.ibasis = torch.from_numpy(ib.astype(np.float32)).to(self.device).contiguous()           # (2*nb, n_fft)
        self.win_sq = torch.from_numpy((win * win).astype(np.float32)).to(self.device)
        fb = self.front.fb[:, :nb].double().cpu().numpy()                                              # (n_mels, nb)
        self.fb_pinv = torch.from_numpy(np.linalg.pinv(fb).astype(np.float32)).to(self.device).contiguous()  # (nb, n_mels)

    # ---- transforms on the C-ABI GEMM -------------------------------------------------------------------------
    def stft(self, y: torch.Tensor) -> torch.Tensor:
        """y (n,) -> (frames, 2, nb) = [Re ; Im], frames = 1 + n // hop (centred, reflect-padded, periodic Hann)."""
        n = y.numel()
        frames = 1 + n // self.hop
        padded = torch.nn.functional.pad(y.view(1, 1, -1), (self.n_fft // 2, self.n_fft // 2), mode="reflect").view(-1).contiguous()
        spec = torch.empty(frames, 2 * self.nb, device=self.device)
        gemm(padded, self.front.basis, spec, frames, 2 * self.nb, self.n_fft, self.hop, self.n_fft, 2 * self.nb)
        return spec.view(frames, 2, self.nb)

    def istft(self, spec: torch.Tensor) -> torch.Tensor:
        """(frames, 2, nb) -> (hop * (frames - 1),) : windowed inverse DFT rows, overlap-add, window-envelope normalisation."""
        frames = spec.shape[0]
        yf = torch.empty(frames, self.n_fft, device=self.device)
        gemm(spec.contiguous(), self.ibasis, yf, frames, self.n_fft, 2 * self.nb, 2 * self.nb, self.n_fft, self.n_fft, b_k=0)
        total = self.n_fft + self.hop * (frames - 1)
        fold = lambda cols: torch.nn.functional.fold(cols, (1, total), (1, self.n_fft), stride=(1, self.hop)).view(-1)
        y = fold(yf.t().reshape(1, self.n_fft, frames))
        env = fold(self.win_sq.view(1, self.n_fft, 1).expand(1, self.n_fft, frames).contiguous())
        y = y / torch.clamp(env, min=1e-8)
        return y[self.n_fft // 2: total - self.n_fft // 2]

    # ---- Griffin-Lim ----------------------------------------------------------------------------------------------
    def magnitude_to_audio(self, S: torch.Tensor, seed: int = 0) -> torch.Tensor:
        """S (frames, nb) linear magnitude -> waveform; the fast Griffin-Lim recursion librosa.griffinlim runs."""
        if self.hop * (S.shape[0] - 1) <= self.n_fft // 2:      # too short to reflect-pad (< 4 frames): silence of that length
            return torch.zeros(self.hop * max(S.shape[0] - 1, 0), device=self.device)
        g = torch.Generator(device="cpu").manual_seed(seed)
        ph = (2 * np.pi * torch.rand(S.shape, generator=g)).to(self.device)
        ang = torch.stack([torch.cos(ph), torch.sin(ph)], 1)                     # (frames, 2, nb), unit modulus
        S = S.to(self.device, torch.float32)
        tprev = None
        alpha = self.momentum / (1 + self.momentum)
        for _ in range(self.n_iter):
            rebuilt = self.stft(self.istft(ang * S[:, None, :]))
            ang = rebuilt if tprev is None else rebuilt - alpha * tprev
            ang = ang / (torch.sqrt((ang * ang).sum(1, keepdim=True)) + 1e-16)
            tprev = rebuilt
        return self.istft(ang * S[:, None, :])

    def mel_to_linear(self, mel_mag: torch.Tensor) -> torch.Tensor:
        """(frames, n_mels) mel magnitude -> (frames, nb) linear magnitude (clamped minimum-norm solution)."""
        m = mel_mag.to(self.device, torch.float32).contiguous()
        out = torch.empty(m.shape[0], self.nb, device=self.device)
        gemm(m, self.fb_pinv, out, m.shape[0], self.nb, m.shape[1], m.shape[1], m.shape[1], self.nb)
        return torch.clamp_(out, min=0.0)

    def mel_to_audio(self, log_mel: torch.Tensor, seed: int = 0) -> torch.Tensor:
        """(frames, n_mels) natural-log mel (the model's output) -> waveform in [-1, 1] scale of the input speech."""
        return self.magnitude_to_audio(self.mel_to_linear(torch.exp(log_mel)), seed)


def write_wav(path: str, y: torch.Tensor, sample_rate: int):
    """16-bit PCM through the stdlib (soundfile is absent here); peaks above full scale are normalised down."""
    x = y.detach().float().cpu().numpy()
    peak = float(np.max(np.abs(x))) if x.size else 0.0
    if peak > 1.0:
        x = x / peak
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(sample_rate))
        w.writeframes(pcm.tobytes())
