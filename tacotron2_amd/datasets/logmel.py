"""Device log-mel front-end (the product path for datasets/tts_dataset.py:204).  Builds the window-folded DFT basis and
the slaney mel filterbank once on the host (float64 -> fp32) and runs t2_logmel_fwd."""
from __future__ import annotations

import math

import numpy as np
import torch

import ctypes

from .._lib import call


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


class TacotronMelSpectrogram:
    """Callable with the reference's call shape `mel = self.melspectrogram(wav, id=...)` -> (frames, n_mels) log-mel."""

    def __init__(self, n_mels: int = 80, sample_rate: int = 22050, n_fft: int = 1024, hop_length: int = 256,
                 f_min: float = 0.0, f_max: float = 8000.0, device="cuda:0"):
        self.n_mels, self.sr, self.n_fft, self.hop = n_mels, sample_rate, n_fft, hop_length
        self.device = torch.device(device)
        nb = n_fft // 2 + 1
        self.ldm = (nb + 3) // 4 * 4
        n = np.arange(n_fft)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
        ang = 2 * np.pi * np.outer(np.arange(nb), n) / n_fft
        basis = np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], 0)          # (2*nb, n_fft)
        all_freqs = np.linspace(0, sample_rate // 2, nb)
        f_pts = _mel_to_hz(np.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2))
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts[None, :] - all_freqs[:, None]
        fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
        fbp = np.zeros((n_mels, self.ldm))
        fbp[:, :nb] = fb.T
        self.basis = torch.from_numpy(basis.astype(np.float32)).to(self.device).contiguous()
        self.fb = torch.from_numpy(fbp.astype(np.float32)).to(self.device).contiguous()

    def __call__(self, wav: torch.Tensor, id=None) -> torch.Tensor:
        wav = wav.to(self.device, torch.float32).contiguous()
        n = wav.numel()
        frames = 1 + n // self.hop
        nb = self.n_fft // 2 + 1
        padded = torch.empty(n + self.n_fft, device=self.device)
        spec = torch.empty(frames, 2 * nb, device=self.device)
        mag = torch.empty(frames, self.ldm, device=self.device)
        out = torch.empty(frames, self.n_mels, device=self.device)
        call("t2_logmel_fwd", wav, n, self.basis, self.fb, padded, spec, mag, out, self.n_fft, self.hop, self.n_mels,
             torch.cuda.current_stream().cuda_stream)
        return out

    def batch(self, wavs: torch.Tensor, n: torch.Tensor, n_max: int, T_out: int = 0, want_gate: bool = True):
        """One training batch in one pass (t2_logmel_batch_fwd): wavs (B, ld) fp32 DEVICE rows (zero-filled behind each utterance),
        n (B,) int64 DEVICE sample counts, n_max = their maximum as a HOST integer (the caller decoded the files, it knows).  Returns
        mel (B, T, n_mels) zero-padded, gate (B, T, 1) (ones, last valid frame 0, zero padding) and mel_len (B,) int32 - all on
        the device, T = max(T_out, 1 + n_max // hop).  Enqueued on the current stream; nothing is read back."""
        assert wavs.is_cuda and wavs.dtype == torch.float32 and wavs.dim() == 2 and wavs.stride(1) == 1
        assert n.is_cuda and n.dtype == torch.int64 and n.numel() == wavs.shape[0]
        B = wavs.shape[0]
        T = max(int(T_out), 1 + int(n_max) // self.hop)
        sizes = (ctypes.c_int64 * 5)()
        call("t2_logmel_batch_workspace", B, int(n_max), self.n_fft, self.hop, self.n_mels, sizes)
        ws = getattr(self, "_batch_ws", None)
        if ws is None or any(w.numel() < need for w, need in zip(ws, sizes[:4])):
            ws = [torch.empty(int(need * 1.25) + 1024, device=self.device) for need in sizes[:4]]
            self._batch_ws = ws
        mel = torch.empty(B, T, self.n_mels, device=self.device)
        gate = torch.empty(B, T, 1, device=self.device) if want_gate else None
        mel_len = torch.empty(B, dtype=torch.int32, device=self.device)
        call("t2_logmel_batch_fwd", wavs, wavs.stride(0), n, B, int(n_max), self.basis, self.fb, ws[0], ws[1], ws[2], ws[3], mel, T,
             gate, mel_len, self.n_fft, self.hop, self.n_mels, torch.cuda.current_stream().cuda_stream)
        return mel, gate, mel_len
