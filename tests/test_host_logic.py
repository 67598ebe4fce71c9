"""CPU-only host logic: text -> ids, collate layout, config aliasing, synthetic batches, LR schedule."""
import json

import numpy as np
import torch

from tacotron2_amd.datasets.text import TextEncoder, expand_abbreviations
from tacotron2_amd.run.common import load_config, model_kwargs
from tacotron2_amd.synthetic import ljspeech_batch

ALLOWED = "!'(),.:;? \\-abcdefghijklmnopqrstuvwxyz"


def test_text_to_ids_matches_reference_table():
    enc = TextEncoder(ALLOWED, "^")
    assert enc.num_chars == 39
    assert enc.table[" "] == 1 and enc.table["^"] == 13 and enc.table["a"] == 14 and enc.table["z"] == 39 and enc.table["\\"] == 12
    want = [21, 18, 25, 25, 28, 6, 1, 26, 31, 8, 1, 32, 26, 22, 33, 21, 7, 23, 28, 27, 18, 32, 2, 13]
    assert enc.encode("Hello, Mr. Smith-Jones!") == want
    assert enc.clean("Café 42 — déjà vu") == "cafe   deja vu^"
    assert expand_abbreviations("mrs. smith and dr. who at ft. knox") == "misess smith and doctor who at fort knox"
    assert TextEncoder(ALLOWED, "^", expand_abbrev=True).clean("Mr. X") == "mister x^"


def test_collate_layout():
    from tacotron2_amd.datasets.tts_dataset import collate
    items = []
    for L, T in ((5, 7), (3, 4)):
        gate = torch.ones(T, 1); gate[-1] = 0
        items.append(({"chars_idx": torch.arange(1, L + 1), "mel_spectrogram": torch.randn(T, 80), "gate": gate},
                      {"chars_idx_len": torch.tensor([L]), "mel_spectrogram_len": torch.IntTensor([T]),
                       "speaker_id": torch.IntTensor([2])}, {"text": "x"}))
    d, m, e = collate(items)
    assert d["chars_idx"].shape == (2, 5) and d["chars_idx"][1, 3:].sum() == 0
    assert d["mel_spectrogram"].shape == (2, 7, 80) and float(d["mel_spectrogram"][1, 4:].abs().sum()) == 0.0
    assert d["gate"].shape == (2, 7, 1) and m["chars_idx_len"].tolist() == [5, 3] and m["mel_spectrogram_len"].dtype == torch.int32
    assert m["speaker_id"].shape == (2,) and e["text"] == ["x", "x"]


def test_stale_reference_configs_load(tmp_path):
    """vanilla-* configs of the reference carry `char_embedding_dim` and lack extensions.descriptions."""
    cfg = {"dataset": {"train": "t.csv", "val": "v.csv", "preprocessing": {"allowed_chars": ALLOWED, "end_token": "^", "num_mels": 80}},
           "training": {"lr": 1e-3, "batch_size": 64, "weight_decay": 1e-6, "name": "x", "args": {"max_steps": 100000}},
           "model": {"scheduler_milestones": [0.5, 0.75],
                     "args": {"prenet_dim": 256, "att_rnn_dim": 1024, "att_dim": 128, "rnn_hidden_dim": 1024, "postnet_dim": 512,
                              "dropout": 0.5, "char_embedding_dim": 512, "encoder_kernel_size": 5}},
           "extensions": {"speaker_tokens": {"active": True, "num_speakers": 4}, "controls": {"active": False}}}
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    kw = model_kwargs(load_config(str(p)))
    assert kw["encoded_dim"] == 512 and "char_embedding_dim" not in kw
    assert kw["num_chars"] == 39 and kw["speaker_tokens"] and kw["num_speakers"] == 4
    assert kw["scheduler_milestones"] == [50000, 75000]
    kw2 = model_kwargs(load_config("config/ljspeech_b32.json"))
    assert kw2["encoded_dim"] == 512 and kw2["num_speakers"] == 4


def test_synthetic_batch_shapes():
    b = ljspeech_batch(32, seed=1234, num_speakers=4)
    L, T = b["chars_idx"].shape[1], b["mel_spectrogram"].shape[1]
    assert 13 <= int(b["chars_idx_len"].min()) and L <= 188 and T <= 872 and b["mel_spectrogram_len"].dtype == torch.int32
    for i in range(32):
        l, t = int(b["chars_idx_len"][i]), int(b["mel_spectrogram_len"][i])
        assert int(b["chars_idx"][i, l - 1]) == 13 and b["chars_idx"][i, l:].sum() == 0
        assert float(b["gate"][i, t - 1]) == 0.0 and float(b["gate"][i, :t - 1].min()) == 1.0
    assert float(b["mel_spectrogram"].max()) <= 2.0 and float(b["mel_spectrogram"][b["mel_spectrogram"] != 0].min()) >= np.log(1e-5) - 1e-6


def test_multistep_lr_schedule_matches_torch():
    from tacotron2_amd.trainer import Trainer
    t = Trainer.__new__(Trainer)
    t.base_lr, t.milestones = 1e-3, [3, 5]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[3, 5], gamma=0.1)
    for step in range(8):
        assert abs(t.lr_at(step) - opt.param_groups[0]["lr"]) < 1e-12
        opt.step(); sch.step()


def test_trim_and_wav_roundtrip(tmp_path):
    import wave
    from tacotron2_amd.datasets.tts_dataset import load_wav, trim_silence
    sr = 22050
    x = np.zeros(sr, np.float32); x[6000:14000] = 0.5 * np.sin(np.arange(8000) / 5.0)
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
    y, r = load_wav(str(tmp_path / "a.wav"))
    assert r == sr and np.abs(y - x).max() < 1e-4
    tr = trim_silence(y, top_db=40)
    assert 7000 < len(tr) < 12000


def test_length_bucket_sampler_covers_every_utterance_and_cuts_padding():
    """LengthBucketBatchSampler: a permutation of the data set every epoch (minus the dropped tail), different between epochs,
    and a much smaller padded/valid ratio than random batches on LJSpeech-shaped lengths (SURVEY.md section 8d distribution)."""
    import numpy as np
    from tacotron2_amd.datasets.tts_dataset import LengthBucketBatchSampler
    rng = np.random.default_rng(0)
    lens = np.clip(np.round(rng.normal(101, 33.6, 4000)), 13, 188).astype(int).tolist()
    s = LengthBucketBatchSampler(lens, batch_size=32, window=16, drop_last=True, seed=1)
    e1, e2 = list(s), list(s)
    flat = [i for b in e1 for i in b]
    assert len(e1) == len(s) == 125 and len(set(flat)) == len(flat) == 4000 and all(len(b) == 32 for b in e1)
    assert e1 != e2
    ratio = lambda batches: sum(max(lens[i] for i in b) * len(b) for b in batches) / sum(lens[i] for b in batches for i in b)
    perm = rng.permutation(4000).tolist()
    random_batches = [perm[i:i + 32] for i in range(0, 4000, 32)]
    assert ratio(random_batches) > 1.45 and ratio(e1) < 1.12, (ratio(random_batches), ratio(e1))


def test_mel_filterbank_known_answers_of_the_slaney_scale():
    """The product's mel filterbank (tacotron2_amd/datasets/logmel.py) against facts that do not come from its own formula or
    from oracle/logmel_ref.py (which restates the same definition): the published constants of the Slaney / Auditory-Toolbox
    scale that torchaudio's mel_scale="slaney" implements (linear below 1 kHz at 200/3 Hz per mel, so 1 kHz = mel 15; above it
    27 mels per factor 6.4) and the properties of norm="slaney" triangles (unit area, peak at the centre frequency, support
    between the neighbouring centres).  The reference's own numbers stay unpinned (speech_utils / torchaudio absent)."""
    import math
    from tacotron2_amd.datasets.logmel import TacotronMelSpectrogram
    fe = TacotronMelSpectrogram(device="cpu")
    nb = 513
    fb = fe.fb[:, :nb].double().numpy()
    assert fb.shape == (80, nb) and float(fe.fb[:, nb:].abs().max()) == 0.0 and (fb >= 0).all()
    df = 11025.0 / (nb - 1)                                   # torchaudio: all_freqs = linspace(0, sr // 2, n_freqs)
    freqs = np.arange(nb) * df
    # centre frequencies from the published scale: 82 points equally spaced in mel between 0 Hz (mel 0) and 8 kHz
    mel_8k = 15.0 + 27.0 * math.log(8.0) / math.log(6.4)      # = 45.2457...
    assert abs(mel_8k - 45.2457) < 1e-3
    m = np.linspace(0.0, mel_8k, 82)
    f = np.where(m < 15.0, m * 200.0 / 3.0, 1000.0 * np.exp((m - 15.0) * math.log(6.4) / 27.0))
    assert abs(f[-1] - 8000.0) < 1e-6
    for k in range(80):
        lo, c, hi = f[k], f[k + 1], f[k + 2]
        row = fb[k]
        nz = np.nonzero(row)[0]
        assert freqs[nz[0]] > lo - 1e-9 and freqs[nz[-1]] < hi + 1e-9, k            # support between the neighbouring centres
        assert abs(freqs[int(row.argmax())] - c) <= df, k                            # peak at the centre (to one bin)
        peak = 2.0 / (hi - lo)                                                       # unit-area triangle of base hi - lo
        inside = (freqs > lo) & (freqs < hi)
        tri = np.where(freqs <= c, (freqs - lo) / (c - lo), (hi - freqs) / (hi - c)) * peak
        assert np.allclose(row[inside], tri[inside], rtol=1e-6, atol=1e-12), k        # exact triangle values at the bins
    wide = [k for k in range(80) if f[k + 2] - f[k] > 12 * df]
    areas = fb[wide].sum(1) * df
    assert len(wide) > 20 and np.abs(areas - 1.0).max() < 0.02                        # unit area once a filter spans enough bins
    assert float(fb[:, freqs > 8000.0 + df].max()) == 0.0                            # nothing above f_max


def test_pipeline_chunk_sizes_cover_the_frames_and_shrink_at_the_end():
    """engine._chunk_sizes: every frame in exactly one chunk, equal chunks except the ramp CH/2, CH/4, CH/8, CH/8 over the last CH
    frames (the un-overlapped end of the forward pipeline / start of the backward one); no ramp for short loops or tiny chunks."""
    from tacotron2_amd.engine import _chunk_sizes
    for T in (1, 7, 63, 64, 65, 100, 872, 5000):
        for CH in (4, 8, 16, 64, 80):
            for ramp in (True, False):
                z = _chunk_sizes(T, CH, ramp_at_end=ramp)
                assert sum(z) == T and all(0 < n <= CH for n in z), (T, CH, ramp, z)
    assert _chunk_sizes(872, 64) == [64] * 12 + [40, 32, 16, 8, 8]
    assert _chunk_sizes(872, 80)[-4:] == [40, 20, 10, 10]
    assert _chunk_sizes(872, 64, ramp_at_end=False) == [64] * 13 + [40]
    assert _chunk_sizes(37, 8) == [8, 8, 8, 8, 5]            # chunks below 16 frames are not ramped


def test_length_bucket_sampler_shards_similar_lengths_across_ranks():
    """Data-parallel batches: all ranks draw one permutation, rank r takes the r-th slice of every sorted super-batch.  The
    ranks see disjoint utterances, run the same number of steps, and at every step their batches have similar lengths - the
    step's global padding (Trainer.global_pad pads every shard to the longest batch of the step) stays near the single-rank
    bucketed ratio instead of falling back to the ratio of random batches."""
    from tacotron2_amd.datasets.tts_dataset import LengthBucketBatchSampler
    rng = np.random.default_rng(0)
    lengths = np.clip(np.round(rng.normal(101, 33.6, 13100)), 13, 188).astype(int).tolist()
    world, bs = 8, 32
    per_rank = [list(LengthBucketBatchSampler(lengths, bs, window=16, seed=0, rank=r, world=world)) for r in range(world)]
    steps = len(per_rank[0])
    assert steps == 13100 // (bs * world) and all(len(p) == steps for p in per_rank)
    seen = [i for p in per_rank for b in p for i in b]
    assert len(seen) == len(set(seen)) == steps * bs * world                    # disjoint, every utterance at most once
    L = np.array(lengths)
    padded = sum(max(L[p[s]].max() for p in per_rank) * bs * world for s in range(steps))
    valid = sum(L[p[s]].sum() for p in per_rank for s in range(steps))
    ratio = padded / valid
    # independent per-rank buckets (seed = rank, the old behaviour): the longest of 8 unrelated batches sets the shape
    indep = [list(LengthBucketBatchSampler(lengths[r::world], bs, window=16, seed=r)) for r in range(world)]
    st2 = min(len(p) for p in indep)
    Ls = [np.array(lengths[r::world]) for r in range(world)]
    padded2 = sum(max(Ls[r][indep[r][s]].max() for r in range(world)) * bs * world for s in range(st2))
    valid2 = sum(Ls[r][indep[r][s]].sum() for r in range(world) for s in range(st2))
    assert ratio < 1.12 and ratio < 0.8 * (padded2 / valid2), (ratio, padded2 / valid2)
    # one rank: unchanged behaviour (every utterance once, drop_last)
    one = list(LengthBucketBatchSampler(lengths, bs, window=16, seed=0))
    assert len(one) == 13100 // bs and len({i for b in one for i in b}) == len(one) * bs


def _wav_files(tmp_path, lengths, sr=22050):
    import wave
    rng = np.random.default_rng(3)
    names = []
    for i, n in enumerate(lengths):
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        x = np.concatenate([np.zeros(2500, np.float32), x, np.zeros(1800, np.float32)])
        with wave.open(str(tmp_path / f"u{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        names.append(f"u{i}.wav")
    return names


def test_trim_silence_running_sum_equals_the_framed_definition():
    """librosa.effects.trim semantics (datasets/tts_dataset.py:197-199 of the reference; parity unpinned - librosa is absent): the O(n)
    running-sum form against the literal (frames x 2048) gather it replaced, on signals with silent heads and tails."""
    from tacotron2_amd.datasets.tts_dataset import trim_silence

    def framed(x, top_db=60, frame_length=2048, hop_length=512):
        if len(x) < frame_length:
            return x
        xp = np.pad(x, (frame_length // 2, frame_length // 2))
        nfr = 1 + (len(xp) - frame_length) // hop_length
        idx = np.arange(frame_length)[None, :] + hop_length * np.arange(nfr)[:, None]
        rms = np.sqrt(np.mean(xp[idx] ** 2, axis=1))
        db = 20 * np.log10(np.maximum(rms, 1e-10)) - 20 * np.log10(max(rms.max(), 1e-10))
        nz = np.nonzero(db > -top_db)[0]
        return x[:0] if len(nz) == 0 else x[nz[0] * hop_length:min(len(x), (nz[-1] + 1) * hop_length)]
    rng = np.random.default_rng(0)
    for k in range(25):
        n = int(rng.integers(2048, 120000))
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        a, b = int(rng.integers(0, n // 3)), int(rng.integers(0, n // 3))
        x[:a] *= 1e-5; x[n - b:] *= 1e-5
        want, got = framed(x), trim_silence(x)
        assert len(want) == len(got) and (len(got) == 0 or (got[0] == want[0] and got[-1] == want[-1])), k
    # digital silence has no peak to be 60 dB below: every frame is at 0 dB of the floor and the signal is kept whole (as librosa does)
    assert len(trim_silence(np.zeros(5000, np.float32))) == 5000 and len(trim_silence(np.ones(100, np.float32))) == 100


def test_device_batch_loader_host_side_packs_one_buffer_and_knows_the_shape(tmp_path):
    """The host half of the training input pipeline (DeviceBatchLoader.host_batch -> HostWavBatch; no GPU needed): the batch's WAVs
    decoded by the thread pool, trimmed, silence-padded and packed into ONE zero-filled buffer; the padded lengths (L, T) as host
    integers - frames = 1 + samples // 256, what the device pass will produce - so a data-parallel step's shape can be agreed
    before anything is on the device (Trainer.negotiate_collated -> set_global_shape)."""
    from tacotron2_amd.datasets.tts_dataset import DeviceBatchLoader, TTSDataset
    lengths = [9000, 15000, 12345, 30000, 7000]
    files = _wav_files(tmp_path, lengths)
    texts = ["short.", "a somewhat longer sentence.", "mid one", "the longest utterance of them all, by far.", "x"]
    ds = TTSDataset(filenames=files, texts=texts, base_dir=str(tmp_path), speaker_ids=[0, 1, 2, 3, 0], silence=512, trim=True, cache=False,
                    device="cpu")
    ld = DeviceBatchLoader(ds, batch_size=2, shuffle=False, drop_last=True, decode_threads=2)
    assert len(ld) == 2
    got = list(ld)
    assert [hb.idxs for hb in got] == [[0, 1], [2, 3]]
    hb = got[1]
    for r, i in enumerate(hb.idxs):
        one = ds.load_audio(i)                                  # the per-utterance host path: the same samples, then zeros
        assert hb.n[r] == len(one) and abs(len(one) - (lengths[i] + 512)) <= 3072      # trimmed to the signal (within a 2048-window + two hops), + silence
        assert np.array_equal(hb.wav[r, :len(one)].numpy(), one) and float(hb.wav[r, len(one):].abs().sum()) == 0.0
    assert hb.wav.shape[0] == 2 and hb.wav.shape[1] % 64 == 0 and hb.wav.shape[1] >= max(hb.n)
    assert hb.frames == [1 + n // 256 for n in hb.n] and hb.T == max(hb.frames)
    assert hb.L == max(len(ds.ids[i]) for i in hb.idxs) and (hb.Lg, hb.Tg) == (hb.L, hb.T) and not hb.hit_mels
    hb.set_global_shape(hb.L + 4, hb.T + 9)
    assert (hb.Lg, hb.Tg) == (hb.L + 4, hb.T + 9)
    assert ld.batches == 2 and ld.decode_s > 0
    # the same through the data-parallel hook on one rank: a no-op
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    from oracle import tacotron2_ref as R
    from tests.helpers import SMALL
    tr = Trainer(ParamStore(R.default_dims(**SMALL), "cpu"), lr=1e-3, weight_decay=0.0)
    assert tr.negotiate_collated(got[0]) is got[0]
