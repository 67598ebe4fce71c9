"""TTSDataset / collate with the reference's item and batch layout (datasets/tts_dataset.py:184-302,
datasets/tts_dataloader.py:8-35): three dicts (data, metadata, extra); chars (B,L) int64 pad 0, mel (B,T,M) pad 0,
gate (B,T,1) = ones with the last valid frame 0, lengths stacked to (B,).  The log-mel runs on the DEVICE
(tacotron2_amd.datasets.logmel).  PCM WAV decoding uses the stdlib `wave` module (torchaudio/librosa are absent).

Two ways to batches:
  * the reference's: TTSDataset.__getitem__ + collate (validation, `main.py test`, `train-mel-export`) - one item at a time, the
    item's mel computed on the device and LEFT there;
  * training: DeviceBatchLoader - what the reference's 8 DataLoader workers + pin_memory are for (run/train.py:150-158), re-planned:
    the B WAVs of a batch are decoded and trimmed by a small thread pool (`wave` / numpy release the GIL), packed into ONE pinned
    buffer, copied up ONCE, and ONE batched log-mel pass (t2_logmel_batch_fwd) writes the padded (B, T, 80) tensor, the gate and
    the lengths on the device.  Nothing comes back to the host."""
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
    """librosa.effects.trim semantics: keep from the first to the last frame whose RMS is within top_db of the peak
    (centred frames, zero padding).  Frame energies from one running sum of squares (float64): O(n), no (frames x 2048) gather."""
    if len(x) < frame_length:
        return x
    pad = frame_length // 2
    cs = np.zeros(len(x) + 2 * pad + 1, np.float64)
    np.cumsum(np.square(x, dtype=np.float64), out=cs[pad + 1:pad + 1 + len(x)])
    cs[pad + 1 + len(x):] = cs[pad + len(x)]
    nfr = 1 + (len(x) + 2 * pad - frame_length) // hop_length
    i0 = hop_length * np.arange(nfr)
    ms = np.maximum(cs[i0 + frame_length] - cs[i0], 0.0) / frame_length
    rms = np.sqrt(ms)
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

    def cache_writer(self):
        if getattr(self, "_cache_writer", None) is None:
            self._cache_writer = _CacheWriter()
        return self._cache_writer

    def flush_cache(self):
        if getattr(self, "_cache_writer", None) is not None:
            self._cache_writer.flush()

    def cache_path(self, i: int):
        return os.path.join(self.cache_dir, f"{self.filenames[i].replace('/', '_')}.pt") if self.cache else None

    def load_audio(self, i: int) -> np.ndarray:
        """Decoded, trimmed, silence-padded utterance i as contiguous float32 (datasets/tts_dataset.py:191-202): host work only."""
        wav, _ = load_wav(os.path.join(self.base_dir, self.filenames[i]))
        if self.trim:
            wav = trim_silence(wav, self.trim_top_db, self.trim_frame_length)
        if self.silence:
            wav = np.pad(wav, (0, self.silence))
        return np.ascontiguousarray(wav, dtype=np.float32)

    @staticmethod
    def write_cache(cache_path: str, mel_host: torch.Tensor):
        # several readers share one cache directory (validation set, the prefetch thread, N data-parallel ranks): the
        # entry appears under its final name only when complete (same pattern as checkpoint.save_atomic)
        tmp = f"{cache_path}.tmp.{os.getpid()}.{threading.get_ident()}"
        torch.save(mel_host.clone(), tmp)
        os.replace(tmp, cache_path)

    def __getitem__(self, i: int):
        fn = self.filenames[i]
        mel = None
        cache_path = None
        if self.cache:
            cache_path = os.path.join(self.cache_dir, f"{fn.replace('/', '_')}.pt")
            if os.path.exists(cache_path):
                try:        # (items are device-resident either way: a batch may mix cache hits and misses)
                    mel = torch.load(cache_path, weights_only=True).to(self.melspectrogram.device)
                except Exception:       # unreadable entry (e.g. left by a killed run): a cache miss, recomputed below
                    mel = None
        if mel is None:
            wav = self.load_audio(i)
            # (the item's mel is computed on the device and stays there: collate pads device tensors, nothing is copied back)
            mel = self.melspectrogram(torch.from_numpy(wav), id=str(i))
            if cache_path is not None:
                self.write_cache(cache_path, mel.cpu())       # the optional on-disk cache is written from a side copy
        gate = torch.ones(len(mel), 1, device=mel.device)
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


class HostWavBatch:
    """One training batch between the host and the device: decoded audio of the cache misses packed in ONE pinned buffer, cached
    mels of the hits, text ids and metadata - everything the host knows, including the batch's own padded lengths (L, T) as HOST
    integers (frames = 1 + samples // hop: no device value is needed to know the shape).  `set_global_shape` records the
    data-parallel step's global (L, T) (Trainer.negotiate_collated); `to_device` does the copies and the ONE batched log-mel pass."""

    def __init__(self, ds, idxs, wav, n, hit_mels, frames):
        self.ds, self.idxs, self.wav, self.n, self.hit_mels, self.frames = ds, idxs, wav, n, hit_mels, frames
        self.L = max(len(ds.ids[i]) for i in idxs)
        self.T = max(frames)
        self.Lg, self.Tg = self.L, self.T

    def set_global_shape(self, Lg: int, Tg: int):
        assert Lg >= self.L and Tg >= self.T
        self.Lg, self.Tg = int(Lg), int(Tg)

    def to_device(self, dev):
        ds, idxs, B, M = self.ds, self.idxs, len(self.idxs), ds_num_mels(self.ds)
        pin = dev.type == "cuda"
        up = lambda t: t.to(dev, non_blocking=True)
        chars = torch.zeros(B, self.Lg, dtype=torch.int64, pin_memory=pin)
        for b, i in enumerate(idxs):
            chars[b, :len(ds.ids[i])] = ds.ids[i]
        out = dict(chars_idx=up(chars), chars_idx_len=up(torch.tensor([len(ds.ids[i]) for i in idxs], dtype=torch.int64)))
        miss = [b for b in range(B) if b not in self.hit_mels]
        mel = gate = mel_len = None
        if miss:
            n_dev = up(torch.tensor([self.n[b] for b in miss], dtype=torch.int64))
            mel_m, gate_m, len_m = ds.melspectrogram.batch(up(self.wav), n_dev, max(self.n[b] for b in miss), T_out=self.Tg)
            if ds.cache:       # side copy for the on-disk cache: leaves through the writer thread once the copy has landed
                ds.cache_writer().submit([ds.cache_path(idxs[b]) for b in miss], [self.frames[b] for b in miss], mel_m)
            if len(miss) == B:
                mel, gate, mel_len = mel_m, gate_m, len_m
        if mel is None:            # cache hits in the batch: their mels go up from the host, padded, and the batch is assembled
            hit = sorted(self.hit_mels)
            hm = torch.zeros(len(hit), self.Tg, M, pin_memory=pin)
            for r, b in enumerate(hit):
                hm[r, :self.frames[b]] = self.hit_mels[b]
            mel = torch.empty(B, self.Tg, M, device=dev)
            mel[torch.tensor(hit, device=dev)] = up(hm)
            if miss:
                mel[torch.tensor(miss, device=dev)] = mel_m
            mel_len = up(torch.tensor(self.frames, dtype=torch.int32))
            t = torch.arange(self.Tg, device=dev)[None, :, None]
            gate = (t < (mel_len.to(torch.int64)[:, None, None] - 1)).float()
        out.update(mel_spectrogram=mel, gate=gate, mel_spectrogram_len=mel_len)
        if ds.speaker_ids is not None:
            out["speaker_id"] = up(torch.tensor([ds.speaker_ids[i] for i in idxs], dtype=torch.int32))
        if ds.description_embeddings is not None:
            rows = []
            for i in idxs:
                p = ds.description_embeddings[i]
                rows.append(torch.load(os.path.join(ds.base_dir, p), map_location="cpu", weights_only=True).view(-1).float()
                            if p is not None else torch.zeros(ds.description_embeddings_dim))
            out["description_embeddings"] = up(torch.stack(rows))
        if ds.features is not None:
            out["controls"] = up(torch.tensor([ds.features[i] for i in idxs], dtype=torch.float32))
        return out


def ds_num_mels(ds) -> int:
    return ds.melspectrogram.n_mels


class _CacheWriter:
    """Writes mel-cache entries from a side copy: the loader thread enqueues a device->pinned-host copy behind the batched log-mel
    and goes on; this thread waits for the copy's event and saves one file per utterance (atomic rename)."""

    def __init__(self):
        import queue
        self.q = queue.Queue()
        self.th = threading.Thread(target=self._work, daemon=True)
        self.th.start()

    def submit(self, paths, frames, mel_dev):
        host = torch.empty(mel_dev.shape, dtype=mel_dev.dtype, pin_memory=True)
        host.copy_(mel_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(mel_dev.device))
        self.q.put((paths, frames, host, ev))

    def _work(self):
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                paths, frames, host, ev = item
                ev.synchronize()
                for r, (p, f) in enumerate(zip(paths, frames)):
                    if not os.path.exists(p):
                        TTSDataset.write_cache(p, host[r, :f])
            finally:
                self.q.task_done()

    def flush(self):
        self.q.join()


class DeviceBatchLoader:
    """Training batches, assembled for the device (see the module docstring).  Iterating yields HostWavBatch objects in the
    sampler's order; DevicePrefetcher(loader, lambda b, dev: b.to_device(dev), ...) moves them up a step ahead of the training loop.
    Sampling as TTSDataLoader: shuffled batches of `batch_size` (drop_last), or length buckets (`bucket_window`, sharded over ranks)."""

    def __init__(self, dataset, batch_size: int, shuffle: bool = True, drop_last: bool = True, bucket_window: int = 0, seed: int = 0,
                 rank: int = 0, world: int = 1, decode_threads: int = 4):
        from concurrent.futures import ThreadPoolExecutor
        self.ds, self.batch_size = dataset, batch_size
        if bucket_window and batch_size > 1:
            self.sampler = LengthBucketBatchSampler([len(i) for i in dataset.ids], batch_size, bucket_window, drop_last, seed, rank, world)
        else:
            g = torch.Generator().manual_seed(seed)
            base = torch.utils.data.RandomSampler(dataset, generator=g) if shuffle else torch.utils.data.SequentialSampler(dataset)
            self.sampler = torch.utils.data.BatchSampler(base, batch_size, drop_last)
        self.pool = ThreadPoolExecutor(max_workers=max(1, decode_threads))
        self.decode_s = 0.0           # host seconds spent decoding / packing (all batches so far), for the throughput record
        self.batches = 0

    def __len__(self):
        return len(self.sampler)

    def _load(self, i):
        ds = self.ds
        if ds.cache:
            cp = ds.cache_path(i)
            if os.path.exists(cp):
                try:
                    return torch.load(cp, weights_only=True)
                except Exception:       # unreadable entry (e.g. left by a killed run): a cache miss
                    pass
        return ds.load_audio(i)

    def host_batch(self, idxs) -> HostWavBatch:
        import time
        t0 = time.perf_counter()
        ds, hop = self.ds, self.ds.melspectrogram.hop
        items = list(self.pool.map(self._load, idxs))
        hit_mels = {b: x for b, x in enumerate(items) if torch.is_tensor(x)}
        miss = [b for b in range(len(idxs)) if b not in hit_mels]
        n = [0 if b in hit_mels else len(items[b]) for b in range(len(idxs))]
        frames = [len(hit_mels[b]) if b in hit_mels else 1 + n[b] // hop for b in range(len(idxs))]
        wav = None
        if miss:
            short = [idxs[b] for b in miss if n[b] <= ds.melspectrogram.n_fft // 2]
            assert not short, f"utterances shorter than half an analysis window (reflect padding needs more): {short}"
            ld = (max(n) + 63) // 64 * 64
            wav = torch.zeros(len(miss), ld, pin_memory=torch.cuda.is_available())
            w = wav.numpy()
            for r, b in enumerate(miss):
                w[r, :n[b]] = items[b]
        self.decode_s += time.perf_counter() - t0
        self.batches += 1
        return HostWavBatch(ds, list(idxs), wav, n, hit_mels, frames)

    def __iter__(self):
        for idxs in self.sampler:
            yield self.host_batch(idxs)


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
