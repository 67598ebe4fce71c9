"""do_test (run/test.py:29-227): synthesise every utterance of the test manifest and write `<i>.wav` files.

The reference runs `trainer.predict` over a batch-8 loader (which also decodes every test wav although the prediction only
reads the text), vocodes each batch with HiFi-GAN (`generator(mel_post.swapaxes(1, 2))`, wav cut at `mel_lengths * 256`) or
with librosa's Griffin-Lim (`mel_post[:mel_length]`), and logs utterances without a stop to `failures.csv`.  Here the texts
go straight from the manifest through the batched decode path (`Tacotron2.forward(teacher_forcing=False)` on the GPU, any
batch size up to 64 per decode group), and the vocoders are the GEMM-kernel ones of this package (hifigan.py, vocoder.py).

Rules kept from the reference: `|`-separated manifest with QUOTE_NONE (:76-78), the `force_speaker` filter and its two
consistency checks (:81-100), controls from the manifest's feature columns (:102-106), `max_len_override` = 5000 (:147),
`mel_lengths = (gate < 0).argmax` - the first masked frame, 0 when the utterance never stopped (:172,205) -, numbering from
1 in manifest order, `failures.csv` rows `i|text` (:187-193,223-227), 22050 Hz output; see synthesize_manifest for the
vocoding details.  `main.py test-correlation` (run/test_correlation.py) runs the same loop under 51 control-vector overrides.
"""
from __future__ import annotations

import csv
import datetime
import os
from typing import List, Optional

import numpy as np
import torch

from ..datasets.text import TextEncoder
from ..model.tts_model import TTSModel
from .common import model_kwargs


def load_test_model(dataset_config, training_config, model_config, extensions_config, checkpoint, dev, random_seed=None):
    """TTSModel.load_from_checkpoint as run/test.py:116-130 calls it (max_len_override 5000, no scheduler), in eval mode."""
    cfg = dict(dataset=dataset_config, training=training_config, model=model_config, extensions=extensions_config)
    kw = model_kwargs(cfg)
    kw["scheduler_milestones"] = []
    model = TTSModel.load_from_checkpoint(checkpoint, device=dev, **kw)
    model.eval()
    if random_seed is not None:
        model.tacotron2._seed = int(random_seed)
    return model


def check_force_speaker(extensions_config: dict):
    """run/test.py:81-100 / run/test_correlation.py:93-110."""
    spk_cfg, ctl_cfg = extensions_config["speaker_tokens"], extensions_config.get("controls", {"active": False})
    if "force_speaker" in spk_cfg:
        if spk_cfg["active"]:
            raise Exception("Cannot use speaker tokens with force_speaker parameter!")
        if ctl_cfg.get("active") and not all("speaker_norm" in x for x in ctl_cfg["features"]):
            raise Exception("If force_speaker, all controls must be for speaker-normalized values!")
        return True
    return False


def make_vocoders(hifi_gan_checkpoint, pre, dev):
    """(HiFi-GAN generator or None, Griffin-Lim or None, sample rate)."""
    from ..vocoder import GriffinLim
    sr = int(pre.get("sample_rate", 22050))
    if hifi_gan_checkpoint is not None:
        from ..hifigan import Generator
        return Generator.from_checkpoint(hifi_gan_checkpoint, device=dev), None, sr
    return None, GriffinLim(n_mels=int(pre.get("num_mels", 80)), sample_rate=sr, device=dev), sr


def synthesize_manifest(model, df, pre, speech_dir, results_dir, gen, gl, sr, feats, batch_size=8, max_len=5000,
                        random_seed=None, zero_length="test") -> List[str]:
    """The prediction + "Saving WAVs" loops of run/test.py:132-227 and run/test_correlation.py:171-250 over one manifest:
    batches of `batch_size` texts through the batched decode path, `mel_lengths = (gate < 0).argmax` (:172,205), numbering from
    1 in manifest order, `failures.csv` rows `i|text`.

    HiFi-GAN: the generator runs on the whole PADDED row of the batch (masked frames are zeros) and the waveform is cut at
    mel_length * 256, as `generator(mel_post.swapaxes(1, 2))` + `wav[:wav_length]` do in the reference - the generator's
    receptive field spans many frames, so the last samples of an utterance depend on what follows it.
    zero_length: what is written for an utterance that never stopped (mel_length 0): "test" = run/test.py:176-193 (wav_length
    becomes -1 and is applied twice: the row minus its last two samples); "correlation" = run/test_correlation.py:196-209 (an
    empty file).  Both log the utterance.  Griffin-Lim (librosa raises on an empty spectrogram): logged, nothing written."""
    from ..vocoder import write_wav
    dev = model.tacotron2.store.device
    enc = TextEncoder(pre["allowed_chars"], pre.get("end_token"), bool(pre.get("expand_abbreviations", False)))
    texts = [enc.clean(t) for t in df.text]
    ids = [torch.tensor(enc.encode(t), dtype=torch.int64) for t in df.text]
    desc_paths = None
    if model.description_embeddings:
        desc_paths = [None if isinstance(x, float) else x for x in df.description_embedding] \
            if "description_embedding" in df.columns else [None] * len(df)
    os.makedirs(results_dir, exist_ok=True)
    speaker_ids = list(df.speaker_id) if model.speaker_tokens else None
    written: List[str] = []

    def fail(i, text):
        print(f"Error: {i}: {text}")
        with open(os.path.join(results_dir, "failures.csv"), "a") as f:
            f.write(f"{i}|{text}\n")

    i = 0
    for b0 in range(0, len(df), batch_size):
        sel = list(range(b0, min(len(df), b0 + batch_size)))
        chars = torch.nn.utils.rnn.pad_sequence([ids[j] for j in sel], batch_first=True).to(dev)
        lens = torch.tensor([len(ids[j]) for j in sel], dtype=torch.int64, device=dev)
        args = {}
        if speaker_ids is not None:
            args["speaker_id"] = torch.tensor([int(speaker_ids[j]) for j in sel], dtype=torch.int32, device=dev)
        if feats is not None:
            args["controls"] = torch.tensor([feats[j] for j in sel], dtype=torch.float32, device=dev)
        if desc_paths is not None:
            dim = int(model.hparams["description_embeddings_dim"])
            rows = [torch.load(os.path.join(speech_dir, desc_paths[j]), map_location="cpu", weights_only=True).reshape(-1)
                    if desc_paths[j] is not None else torch.zeros(dim) for j in sel]
            args["description_embeddings"] = torch.stack(rows).to(dev)
        with torch.no_grad():
            _, post, gate, _ = model(chars_idx=chars, chars_idx_len=lens, teacher_forcing=False, max_len_override=max_len, **args)
        mel_lengths = (gate[:, :, 0] < 0).to(torch.int64).argmax(dim=-1).cpu().tolist()
        for k, j in enumerate(sel):
            i += 1
            n = int(mel_lengths[k])
            name = os.path.join(results_dir, f"{i}.wav")
            if gen is not None:
                row = gen(post[k].t().contiguous())[0, 0]          # the padded row, as the reference vocodes it
                if n == 0:
                    fail(i, texts[j])
                    wav = row[:-1][:-1] if zero_length == "test" else row[:0]
                else:
                    wav = row[:n * 256]
                write_wav(name, wav.cpu(), sr)
                written.append(name)
            else:
                if n == 0:
                    fail(i, texts[j])
                    continue
                write_wav(name, gl.mel_to_audio(post[k, :n], seed=int(random_seed or 0)), sr)
                written.append(name)
    return written


def do_test(dataset_config: dict, training_config: dict, model_config: dict, extensions_config: dict, device: int,
            speech_dir: Optional[str], checkpoint: str, hifi_gan_checkpoint: Optional[str] = None,
            results_dir: Optional[str] = None, batch_size: int = 8, max_len: int = 5000, limit: Optional[int] = None,
            random_seed: Optional[int] = None) -> List[str]:
    import pandas as pd
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    pre = dataset_config["preprocessing"]
    df = pd.read_csv(dataset_config["test"], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
    if check_force_speaker(extensions_config):
        df = df[df.speaker_id == extensions_config["speaker_tokens"]["force_speaker"]].reset_index(drop=True)
    if limit is not None:
        df = df.iloc[:int(limit)].reset_index(drop=True)
    model = load_test_model(dataset_config, training_config, model_config, extensions_config, checkpoint, dev, random_seed)
    ctl_cfg = extensions_config.get("controls", {"active": False})
    feats = df[ctl_cfg["features"]].values.tolist() if model.controls else None
    if results_dir is None:
        results_dir = f"results_{training_config['name']}_test {datetime.datetime.now()}"
    gen, gl, sr = make_vocoders(hifi_gan_checkpoint, pre, dev)
    return synthesize_manifest(model, df, pre, speech_dir, results_dir, gen, gl, sr, feats, batch_size=batch_size,
                               max_len=max_len, random_seed=random_seed, zero_length="test")
