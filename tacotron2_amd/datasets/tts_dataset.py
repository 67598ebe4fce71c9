"""TTSDataset / collate with the reference's item and batch layout (datasets/tts_dataset.py:184-302,
datasets/tts_dataloader.py:8-35): three dicts (data, metadata, extra); chars (B,L) int64 pad 0, mel (B,T,M) pad 0,
gate (B,T,1) = ones with the last valid frame 0, lengths stacked to (B,).  The log-mel runs on the DEVICE
(tacotron2_amd.datasets.logmel).  PCM WAV decoding uses the stdlib `wave` module (torchaudio/librosa are absent)."""
from __future__ import annotations

import os
import threading
import wave
from collections import defaultdict
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .logmel import TacotronMelSpectrogram
from .text import ALLOWED_CHARS, TextEncoder


def load_wav(path: str):
    with wave.open(path, "rb") as w:
        sr, n, width, ch = w.getframerate(), w.getnframes(), w.getsampwidth(), w.getnchannels()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    if ch > 1:
        x = x.reshape(-1, ch)[:, 0]
    return x, sr


def trim_silence(x: np.ndarray, top_db: float = 60, frame_length: int = 2048, hop_length: int = 512) -> np.ndarray:
    """librosa.effects.trim semantics: keep from the first to the last frame whose RMS is within top_db of the peak."""
    if len(x) < frame_length:
        return x
    pad = frame_length // 2
    xp = np.pad(x, (pad, pad), mode="constant")
    nfr = 1 + (len(xp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(nfr)[:, None]
    rms = np.sqrt(np.mean(xp[idx] ** 2, axis=1))
    db = 20 * np.log10(np.maximum(rms, 1e-10)) - 20 * np.log10(max(rms.max(), 1e-10))
    nz = np.nonzero(db > -top_db)[0]
    if len(nz) == 0:
        return x[:0]
    return x[nz[0] * hop_length:min(len(x), (nz[-1] + 1) * hop_length)]


class TTSDataset(torch.utils.data.Dataset):
    def __init__(self, filenames: List[str], texts: List[str], base_dir: str, speaker_ids: Optional[List[int]] = None,
                 features=None, allowed_chars: str = ALLOWED_CHARS, end_token: Optional[str] = "^", silence: int = 0,
                 trim: bool = True, trim_top_db: int = 60, trim_frame_length: int = 2048, expand_abbreviations=False,
                 include_text=False, include_filename=False, num_mels: int = 80, cache=False, cache_dir=None,
                 description_embeddings: Optional[List[Optional[str]]] = None, description_embeddings_dim: int = 768,
                 sample_rate: int = 22050, device="cuda:0", **_ignored):
        assert (cache and cache_dir is not None) or not cache, "If caching spectrograms, a cache directory is required"
        if cache and not os.path.exists(cache_dir):
            os.makedirs(cache_dir, exist_ok=True)
        self.filenames, self.base_dir = filenames, base_dir
        self.enc = TextEncoder(allowed_chars, end_token, expand_abbreviations)
        self.ids = [torch.tensor(self.enc.encode(t), dtype=torch.int64) for t in texts]
        self.texts = [self.enc.clean(t) for t in texts]
        self.speaker_ids, self.features = speaker_ids, features
        self.silence, self.trim, self.trim_top_db, self.trim_frame_length = silence, trim, trim_top_db, trim_frame_length
        self.cache, self.cache_dir = cache, cache_dir
        self.description_embeddings, self.description_embeddings_dim = description_embeddings, description_embeddings_dim
        self.include_text, self.include_filename = include_text, include_filename
        self.melspectrogram = TacotronMelSpectrogram(n_mels=num_mels, sample_rate=sample_rate, device=device)

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, i: int):
        fn = self.filenames[i]
        mel = None
        cache_path = None
        if self.cache:
            cache_path = os.path.join(self.cache_dir, f"{fn.replace('/', '_')}.pt")
            if os.path.exists(cache_path):
                try:
                    mel = torch.load(cache_path, weights_only=True)
                except Exception:       # unreadable entry (e.g. left by a killed run): a cache miss, recomputed below
                    mel = None
        if mel is None:
            wav, _ = load_wav(os.path.join(self.base_dir, fn))
            if self.trim:
                wav = trim_silence(wav, self.trim_top_db, self.trim_frame_length)
            wav = np.pad(wav, (0, self.silence))
            mel = self.melspectrogram(torch.from_numpy(np.ascontiguousarray(wav)), id=str(i)).cpu()
            if cache_path is not None:
                # several readers share one cache directory (validation set, the prefetch thread, N data-parallel ranks): the
                # entry appears under its final name only when complete (same pattern as checkpoint.save_atomic)
                tmp = f"{cache_path}.tmp.{os.getpid()}.{threading.get_ident()}"
                torch.save(mel, tmp)
                os.replace(tmp, cache_path)
        gate = torch.ones(len(mel), 1)
        gate[-1] = 0.0
        data = {"chars_idx": self.ids[i], "mel_spectrogram": mel, "gate": gate}
        meta = {"chars_idx_len": torch.tensor([len(self.ids[i])], dtype=torch.int64),
                "mel_spectrogram_len": torch.IntTensor([len(mel)]), "gate_len": torch.IntTensor([len(gate)])}
        extra: Dict[str, Any] = {}
        if self.include_text:
            extra["text"] = self.texts[i]
        if self.include_filename:
            extra["filename"] = fn
        if self.speaker_ids is not None:
            meta["speaker_id"] = torch.IntTensor([self.speaker_ids[i]])
        if self.description_embeddings is not None:
            p = self.description_embeddings[i]
            if p is not None:
                meta["description_embeddings"] = torch.load(os.path.join(self.base_dir, p), map_location="cpu",
                                                            weights_only=True).unsqueeze(0)
            else:
                meta["description_embeddings"] = torch.zeros(1, self.description_embeddings_dim)
        if self.features is not None:
            meta["features"] = torch.Tensor([self.features[i]])
        return data, meta, extra


def collate(items):
    """datasets/tts_dataloader.py:8-35."""
    data, meta, extra = defaultdict(list), defaultdict(list), defaultdict(list)
    for d, m, e in items:
        for k, v in d.items():
            data[k].append(v)
        for k, v in m.items():
            meta[k].append(v)
        for k, v in e.items():
            extra[k].append(v)
    out_d = {k: torch.nn.utils.rnn.pad_sequence(v, batch_first=True) for k, v in data.items()}
    out_m = {k: torch.stack(v).squeeze(1) for k, v in meta.items()}
    return out_d, out_m, dict(extra)


class LengthBucketBatchSampler(torch.utils.data.Sampler):
    """Batches of utterances of similar length.  Text length predicts the frame count (correlation 0.957 over the reference
    manifests), so it is known without decoding audio: every epoch the shuffled utterances are cut into
    windows of `window` batches, each window is sorted by text length and cut into batches, and the batches are shuffled.
    At b = 32 the padded-to-valid frame ratio of LJSpeech-shaped data drops from 1.53 (random batches) to about 1.1 - the
    padding, not the 112 MB gradient all-reduce, is what limits data-parallel efficiency.  The model
    sees every utterance once per epoch either way; only the batch composition changes.

    Data parallelism (rank, world): every rank draws the SAME permutation (same seed), each sorted window is cut into
    super-batches of world * batch_size utterances and rank r takes the r-th slice of each.  The ranks then see disjoint data
    and, at every step, batches of similar length - Trainer.global_pad pads all shards to the step's global (L, T), so one rank
    with a long batch would otherwise set the shape for all of them.  (With world > 1 the dataset must hold the WHOLE manifest,
    not a per-rank slice.)"""

    def __init__(self, lengths, batch_size: int, window: int = 16, drop_last: bool = True, seed: int = 0, rank: int = 0,
                 world: int = 1):
        self.lengths, self.batch_size, self.window, self.drop_last = list(lengths), batch_size, max(1, window), drop_last
        self.epoch, self.seed = 0, seed
        assert 0 <= rank < world
        self.rank, self.world = rank, world

    def __len__(self):
        n, sb = len(self.lengths), self.batch_size * self.world
        if self.world > 1:
            return n // sb           # whole super-batches only: every rank runs the same number of steps
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        self.epoch += 1
        perm = torch.randperm(len(self.lengths), generator=g).tolist()
        sb = self.batch_size * self.world
        span = sb * self.window
        if self.world > 1:
            perm = perm[:len(perm) // sb * sb]
        batches = []
        for w0 in range(0, len(perm), span):
            win = sorted(perm[w0:w0 + span], key=lambda i: self.lengths[i])
            for b0 in range(0, len(win), sb):
                b = win[b0 + self.rank * self.batch_size:b0 + (self.rank + 1) * self.batch_size]
                if len(b) == self.batch_size or (not self.drop_last and self.world == 1 and b):
                    batches.append(b)
        for i in torch.randperm(len(batches), generator=g).tolist():      # same order on every rank (same generator state)
            yield batches[i]


def TTSDataLoader(dataset, batch_size=1, num_workers=0, shuffle=None, drop_last=True, bucket_window=0, seed=0, rank=0, world=1,
                  **_ignored):
    """The log-mel runs on the GPU inside __getitem__, so items are produced in-process (num_workers = 0).
    bucket_window > 0: length-bucketed batches (LengthBucketBatchSampler) instead of the reference's plain shuffle; with
    world > 1 the sampler also shards them across the ranks (the dataset then holds the whole manifest)."""
    if bucket_window and batch_size > 1 and hasattr(dataset, "ids"):
        sampler = LengthBucketBatchSampler([len(i) for i in dataset.ids], batch_size, bucket_window, drop_last, seed, rank, world)
        return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, collate_fn=collate, num_workers=0)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, collate_fn=collate if batch_size > 1 else None,
                                       num_workers=0, shuffle=shuffle, drop_last=drop_last)


class DevicePrefetcher:
    """Batches of `loader`, ready on the device `depth` steps ahead of the training loop.

    A background thread walks the loader - wav decode and trim on the host, the log-mel kernels of cache misses, collate -
    and moves each batch to the device (`to_device(batch, dev)`) on ITS OWN HIP stream, so host-side item work and the
    host->device copies of batch k+1 run while the training step of batch k occupies the main stream (the reference gets
    the same overlap from DataLoader worker processes + pin_memory, run/train.py:140-158).  ctypes calls and torch copies
    release the GIL.  The consumer's stream waits for the batch's event and the tensors are re-registered with it
    (`record_stream`), so the caching allocator does not hand their memory back to the copy stream while they are in use.

    Data-parallel training: `negotiate(host_batch) -> host_batch` runs in the loader thread on every batch BEFORE it goes to the
    device (Trainer.negotiate_collated: one tiny MAX all-reduce of the padded lengths over the trainer's host-side group, then the
    padding) - the shape of step k+1 is agreed while step k runs and the training loop never waits for a collective result.
    Every rank must then produce exactly the batches it consumes, in the same count: `limit` = the number of batches the consumer
    will take (the thread stops there, so no rank is left inside a collective its peers never enter), `cycle` = start the loader
    again when it is exhausted (a new epoch: shuffling samplers draw their next permutation) instead of ending the iteration -
    ranks whose shards differ by an utterance have epochs of different length, the k-th batch of every rank still meets the k-th
    batch of the others.  `device` may be the CPU (host-logic tests): no stream, no events.

    With `negotiate` the object is SINGLE-USE and must be consumed to its `limit`: `produced` counts batches when they are queued,
    and a consumer that stops early (break, exception) discards queued batches whose shapes its peers have already agreed on - a
    second iteration would then enter fewer collectives than they expect.  A peer that dies leaves this rank's loader thread inside
    the negotiation's all-reduce until the host-side group's timeout (Trainer(shape_timeout_s=...), 300 s), after which the error
    surfaces in the consumer."""

    _END = object()

    def __init__(self, loader, to_device, device, depth: int = 2, negotiate=None, limit: Optional[int] = None, cycle: bool = False):
        self.loader, self.to_device, self.device, self.depth = loader, to_device, torch.device(device), max(1, depth)
        self.negotiate, self.limit, self.cycle = negotiate, limit, bool(cycle)
        self.stream = None
        self.produced = 0          # batches handed to the queue so far (all iterations)

    def __len__(self):
        return len(self.loader)

    def _batches(self, stop):
        while True:
            n = 0
            for b in self.loader:
                n += 1
                yield b
            if not self.cycle or n == 0 or stop.is_set():
                return

    def _work(self, q, stop):
        import contextlib
        try:
            cuda = self.device.type == "cuda"
            if cuda:
                torch.cuda.set_device(self.device)
            with (torch.cuda.stream(self.stream) if cuda else contextlib.nullcontext()):
                if self.limit is None or self.produced < self.limit:
                    for b in self._batches(stop):
                        if stop.is_set():
                            return
                        if self.negotiate is not None:
                            b = self.negotiate(b)
                        out = self.to_device(b, self.device)
                        ev = None
                        if cuda:
                            ev = torch.cuda.Event()
                            ev.record(self.stream)
                        q.put((out, ev))
                        self.produced += 1
                        if self.limit is not None and self.produced >= self.limit:
                            break
            q.put((self._END, None))
        except BaseException as e:      # surfaces in the consumer
            q.put((e, None))

    def __iter__(self):
        import queue
        import threading
        cuda = self.device.type == "cuda"
        if self.stream is None and cuda:
            self.stream = torch.cuda.Stream(device=self.device)
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        th = threading.Thread(target=self._work, args=(q, stop), daemon=True)
        th.start()
        try:
            while True:
                out, ev = q.get()
                if out is self._END:
                    return
                if isinstance(out, BaseException):
                    raise out
                if cuda:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for v in (out.values() if isinstance(out, dict) else out):
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
                yield out
        finally:
            stop.set()
            while th.is_alive():        # unblock a producer waiting on the full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)
