"""The device-resident input pipeline of training (reference: datasets/tts_dataset.py:184-214,
datasets/tts_dataloader.py:8-35, run/train.py:150-158): the batched log-mel pass against the per-utterance path (bit-identical) and
against the float64 restatement (oracle/logmel_ref.py; PARITY UNPINNED against the reference: `speech_utils` is not available),
and DeviceBatchLoader against the reference-style item path + collate."""
import os
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_model import _dev  # noqa: E402


def _signals(lengths, sr=22050, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for i, n in enumerate(lengths):
        t = np.arange(n) / sr
        x = 0.3 * np.sin(2 * np.pi * (180 + 37 * i) * t) + 0.15 * np.sin(2 * np.pi * (2500 + 211 * i) * t) * np.exp(-2 * t) + 0.01 * rng.normal(size=n)
        if n > 9000:
            x[4000:4700] = 0.0                                 # digital silence: the 1e-5 clamp
        out.append(x.astype(np.float32))
    return out


def test_batched_logmel_is_bit_identical_to_per_utterance_and_matches_float64():
    from oracle.logmel_ref import logmel
    from tacotron2_amd.datasets.logmel import TacotronMelSpectrogram
    dev = _dev()
    fe = TacotronMelSpectrogram(device=dev)
    lengths = [22050 + 777, 513, 256 * 40, 256 * 40 - 1, 31000, 1024, 145000]      # shortest legal (n_fft/2 + 1), hop multiples, a 6.6 s one
    sig = _signals(lengths)
    B, n_max = len(sig), max(lengths)
    ld = (n_max + 63) // 64 * 64
    wav = torch.zeros(B, ld)
    for b, x in enumerate(sig):
        wav[b, :len(x)] = torch.from_numpy(x)
    # garbage behind the utterances must not matter?  It does by contract (rows are zero-filled): the reflect padding never reads it,
    # but frames are counted from n - put NaN there to prove nothing behind n_b is read
    wavn = wav.clone()
    for b, x in enumerate(sig):
        wavn[b, len(x):] = float("nan")
    n_dev = torch.tensor(lengths, dtype=torch.int64, device=dev)
    T_extra = 1 + n_max // 256 + 5
    mel, gate, mel_len = fe.batch(wavn.to(dev), n_dev, n_max, T_out=T_extra)
    torch.cuda.synchronize()
    assert mel.shape == (B, T_extra, 80) and gate.shape == (B, T_extra, 1) and mel_len.dtype == torch.int32
    assert mel_len.tolist() == [1 + n // 256 for n in lengths]
    assert bool(torch.isfinite(mel).all())
    for b, x in enumerate(sig):
        f = 1 + len(x) // 256
        one = fe(torch.from_numpy(x), id=str(b))                        # the per-utterance path (TTSDataset.__getitem__)
        assert torch.equal(mel[b, :f], one), b                          # bit-identical
        assert float(mel[b, f:].abs().max()) == 0.0 if f < T_extra else True
        g = gate[b, :, 0].cpu()
        assert g[:f - 1].eq(1).all() and g[f - 1:].eq(0).all()
        ref = logmel(x.astype(np.float64))
        got = mel[b, :f].double().cpu().numpy()
        assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 1e-4, b
    # the default output length is the longest utterance's frame count
    mel2, _, _ = fe.batch(wav.to(dev), n_dev, n_max)
    assert mel2.shape[1] == 1 + n_max // 256 and torch.equal(mel2, mel[:, :mel2.shape[1]])


def _manifest(tmp_path, n=7, sr=22050):
    speech = tmp_path / "wavs"
    speech.mkdir()
    lengths = [sr // 2 + 1733 * i for i in range(n)]
    texts = []
    for i, x in enumerate(_signals(lengths, sr, seed=9)):
        x = np.concatenate([np.zeros(3000, np.float32), x, np.zeros(2500, np.float32)])      # leading / trailing silence: trim has work
        with wave.open(str(speech / f"u{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        texts.append(f"Utterance number {i}{', and a few more words' * (i % 3)}.")
    return speech, [f"u{i}.wav" for i in range(n)], texts


@pytest.mark.parametrize("cache", [False, True])
def test_device_batch_loader_equals_the_item_path_and_collate(tmp_path, cache):
    from tacotron2_amd.datasets.tts_dataset import DeviceBatchLoader, TTSDataset, collate
    dev = _dev()
    speech, files, texts = _manifest(tmp_path)
    kw = dict(filenames=files, texts=texts, base_dir=str(speech), speaker_ids=[i % 4 for i in range(len(files))], silence=512, trim=True,
              cache=cache, cache_dir=str(tmp_path / "cache") if cache else None, device=dev)
    ds = TTSDataset(**kw)
    loader = DeviceBatchLoader(ds, batch_size=3, shuffle=False, drop_last=False, decode_threads=3)
    assert len(loader) == 3

    def check(first_pass):
        for hb, idxs in zip(loader, ([0, 1, 2], [3, 4, 5], [6])):
            assert hb.idxs == idxs and (hb.L, hb.T) == (hb.Lg, hb.Tg)
            got = hb.to_device(dev)
            ds_ref = TTSDataset(**dict(kw, cache=False, cache_dir=None))             # the item path, no cache: always recomputed
            data, meta, _ = collate([ds_ref[i] for i in idxs])
            assert torch.equal(got["chars_idx"].cpu(), data["chars_idx"]) and torch.equal(got["chars_idx_len"].cpu(), meta["chars_idx_len"])
            assert got["mel_spectrogram"].is_cuda and data["mel_spectrogram"].is_cuda      # the item path leaves its mels on the device too
            assert torch.equal(got["mel_spectrogram"], data["mel_spectrogram"])            # bit-identical, padding included
            assert torch.equal(got["gate"], data["gate"])
            assert torch.equal(got["mel_spectrogram_len"].cpu(), meta["mel_spectrogram_len"]) and got["mel_spectrogram_len"].dtype == torch.int32
            assert torch.equal(got["speaker_id"].cpu(), meta["speaker_id"])
            assert (hb.T, hb.L) == (data["mel_spectrogram"].shape[1], data["chars_idx"].shape[1])    # host-side shape = device shape
            if cache:
                assert (len(hb.hit_mels) == 0) == first_pass
    check(True)
    if cache:
        ds.flush_cache()
        assert sorted(os.listdir(tmp_path / "cache")) == sorted(f"{f}.pt" for f in files)
        check(False)                                                             # every utterance from the cache now
        os.remove(tmp_path / "cache" / "u4.wav.pt")                              # a batch of hits AND a miss
        hb = loader.host_batch([3, 4, 5])
        assert sorted(hb.hit_mels) == [0, 2]
        got = hb.to_device(dev)
        data, meta, _ = collate([TTSDataset(**dict(kw, cache=False, cache_dir=None))[i] for i in (3, 4, 5)])
        assert torch.equal(got["mel_spectrogram"], data["mel_spectrogram"]) and torch.equal(got["gate"], data["gate"])
    # a data-parallel step's global shape: padding only
    hb = loader.host_batch([0, 1, 2])
    hb.set_global_shape(hb.L + 3, hb.T + 7)
    got = hb.to_device(dev)
    assert got["chars_idx"].shape == (3, hb.L + 3) and got["mel_spectrogram"].shape == (3, hb.T + 7, 80) and got["gate"].shape == (3, hb.T + 7, 1)
    assert float(got["mel_spectrogram"][:, hb.T:].abs().max()) == 0.0 and int(got["chars_idx"][:, hb.L:].sum()) == 0


def test_device_logmel_against_an_fft_based_restatement():
    """An independent route to the same definition (datasets/prosody_dataset.py:39-50,67 of the reference): torch.stft (an FFT, centred,
    reflect padding, periodic Hann) -> magnitude -> the slaney filterbank -> log(clamp(1e-5)).  The device path computes the DFT as a
    GEMM against a window-folded cos/sin basis; oracle/logmel_ref.py uses numpy's FFT on explicitly framed windows.  Three routes, one
    result (still PARITY-UNPINNED against the reference's own `speech_utils`, which is not available)."""
    from oracle.logmel_ref import mel_filterbank
    from tacotron2_amd.datasets.logmel import TacotronMelSpectrogram
    dev = _dev()
    fe = TacotronMelSpectrogram(device=dev)
    x = torch.from_numpy(_signals([50000], seed=21)[0])
    got = fe(x, id="0").double().cpu()
    spec = torch.stft(x.double(), 1024, 256, 1024, torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True,
                      pad_mode="reflect", return_complex=True).abs().T                       # (frames, 513)
    fb = torch.from_numpy(mel_filterbank(22050, 1024, 80, 0.0, 8000.0))                         # (80, 513)
    ref = torch.log(torch.clamp(spec @ fb.T, min=1e-5))
    assert got.shape == ref.shape == (1 + 50000 // 256, 80)
    assert float((got - ref).abs().max()) < 2e-3 and float((got - ref).abs().mean()) < 1e-4
    # and the product's own filterbank is the restatement's (built independently in tacotron2_amd/datasets/logmel.py)
    assert float((fe.fb[:, :513].double().cpu() - fb).abs().max()) < 1e-6
