"""Synthetic LJSpeech-shaped batches: per utterance text length
l ~ clip(round(N(101, 33.6)), 13, 188); frames t = clip(round(5.68*l + N(0, 54)), 99, 872) (reproduces
corr(l, t) = 0.957, mean 570, max 872 of the reference manifests); chars uniform in [1, 39] with the end token
id 13 last; log-mel values ~ N(-5.5, 2.0) clipped to [log 1e-5, 2]; gate target ones with the last valid frame 0
(datasets/tts_dataset.py:213-214); everything zero-padded to the batch maxima (datasets/tts_dataloader.py:25-33)."""
import math

import numpy as np
import torch


def ljspeech_batch(batch: int, seed: int = 1234, num_mels: int = 80, num_speakers: int = 0, desc_dim: int = 0,
                   fixed_shape=None, shape: str = "ljspeech"):
    """shape="libritts": lengths of the descriptions-libritts configuration (BASELINE configs[3]; SURVEY.md section 8d: 24 kHz,
    utterances of at most 10 s).  Fitted to the reference's manifests (data/libritts-train-clean-100.csv joined with
    data/libritts-durations.csv, 27,948 utterances <= 10 s): text length mean 71.5 / std 42.4 / max 238 (right-skewed: a gamma
    distribution with those moments), frames = 1 + 24000 * duration // 256 = 5.31 * l + 16.5 + N(0, 63.4) (corr 0.963),
    16 <= frames <= 938."""
    rng = np.random.default_rng(seed)
    if fixed_shape is not None:
        lens = np.full(batch, fixed_shape[0]); tl = np.full(batch, fixed_shape[1])
    elif shape == "libritts":
        k = (71.5 / 42.4) ** 2
        lens = np.clip(np.round(rng.gamma(k, 71.5 / k, batch)), 2, 238).astype(np.int64)
        tl = np.clip(np.round(5.31 * lens + 16.5 + rng.normal(0, 63.4, batch)), 16, 938).astype(np.int64)
    else:
        lens = np.clip(np.round(rng.normal(101, 33.6, batch)), 13, 188).astype(np.int64)
        tl = np.clip(np.round(5.68 * lens + rng.normal(0, 54, batch)), 99, 872).astype(np.int64)
    L, T = int(lens.max()), int(tl.max())
    chars = np.zeros((batch, L), np.int64)
    mel = np.zeros((batch, T, num_mels), np.float32)
    gate = np.zeros((batch, T, 1), np.float32)
    for b in range(batch):
        chars[b, :lens[b]] = rng.integers(1, 40, lens[b])
        chars[b, lens[b] - 1] = 13
        mel[b, :tl[b]] = np.clip(rng.normal(-5.5, 2.0, (tl[b], num_mels)), math.log(1e-5), 2.0)
        gate[b, :tl[b] - 1] = 1.0
    out = dict(chars_idx=torch.from_numpy(chars), chars_idx_len=torch.from_numpy(lens),
               mel_spectrogram=torch.from_numpy(mel), mel_spectrogram_len=torch.from_numpy(tl.astype(np.int32)),
               gate=torch.from_numpy(gate))
    if num_speakers:
        out["speaker_id"] = torch.from_numpy(rng.integers(0, num_speakers, batch).astype(np.int32))
    if desc_dim:
        out["description_embeddings"] = torch.from_numpy(rng.normal(0, 1, (batch, desc_dim)).astype(np.float32))
    return out
