"""float64 numpy restatement of the log-mel front-end (TEST INFRASTRUCTURE; PARITY UNPINNED: the reference's arithmetic
lives in the un-vendored `speech_utils` package, requirements.txt:12, and no reference file holds its outputs).

Definition from in-repo evidence: torchaudio MelSpectrogram(sample_rate, n_fft=1024, win_length=1024, hop_length=256,
f_min=0, f_max=8000, n_mels=80, power=1, mel_scale="slaney", norm="slaney") then log(clamp(min=1e-5)).T
(datasets/prosody_dataset.py:39-50,67); inverse at synthesis uses the same framing (run/say.py:161-171)."""
import numpy as np


def hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, f_min=0.0, f_max=8000.0):
    """(n_mels, n_fft//2+1) triangular filters, slaney mel scale, slaney area normalisation."""
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0, sr // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel_slaney(f_min), hz_to_mel_slaney(f_max), n_mels + 2)
    f_pts = mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb.T


def logmel(wav, sr=22050, n_fft=1024, hop=256, n_mels=80, f_min=0.0, f_max=8000.0):
    wav = np.asarray(wav, np.float64)
    pad = n_fft // 2
    x = np.pad(wav, (pad, pad), mode="reflect")
    frames = 1 + len(wav) // hop
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)      # periodic Hann
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    spec = np.abs(np.fft.rfft(x[idx] * win[None, :], axis=1))
    mel = spec @ mel_filterbank(sr, n_fft, n_mels, f_min, f_max).T
    return np.log(np.maximum(mel, 1e-5))
