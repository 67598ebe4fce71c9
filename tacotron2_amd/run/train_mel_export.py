"""do_train_mel_export (run/train_mel_export.py:16-142): teacher-forced, eval-mode predictions of the post-net mel for every
utterance of the train and validation manifests, written as `<results>/<wav name with / -> _>.np.npy` (np.save of a
`.np` name) cut to the utterance's true frame count - the ground-truth-aligned mels a HiFi-GAN fine-tune trains on.

Kept from the reference: `|`-separated manifests with QUOTE_NONE, no mel cache (:25), batches of 64 in manifest order with
the last partial batch included (:71-88), speaker ids / controls from the manifest when the model uses them (:117-123),
`mel_out[:mel_spectrogram_len]` per utterance (:137-142).  The batches come through DevicePrefetcher (wav decode, device
log-mel and copies of the next batch overlap the forward of the current one)."""
from __future__ import annotations

import csv
import datetime
import os
from typing import List, Optional

import numpy as np
import torch

from ..model.tts_model import TTSModel
from .common import model_kwargs


def do_train_mel_export(dataset_config: dict, training_config: dict, model_config: dict, extensions_config: dict, device: int,
                        speech_dir: str, checkpoint: str, results_dir: Optional[str] = None, batch_size: int = 64) -> List[str]:
    import pandas as pd
    from ..datasets.tts_dataset import DevicePrefetcher, TTSDataLoader, TTSDataset
    from .train import _to_dev
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    cfg = dict(dataset=dataset_config, training=training_config, model=model_config, extensions=extensions_config)
    kw = model_kwargs(cfg)
    model = TTSModel.load_from_checkpoint(checkpoint, device=dev, **kw)
    model.eval()
    if results_dir is None:
        results_dir = f"results_{training_config['name']}_train_mel_export {datetime.datetime.now()}"
    os.makedirs(results_dir, exist_ok=True)
    pre = dict(dataset_config["preprocessing"])
    pre["cache"] = False
    ctl = extensions_config.get("controls", {"active": False})
    written: List[str] = []
    for split in ("train", "val"):
        df = pd.read_csv(dataset_config[split], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
        ds = TTSDataset(filenames=list(df.wav), texts=list(df.text), base_dir=speech_dir,
                        speaker_ids=list(df.speaker_id) if model.speaker_tokens else None,
                        features=df[ctl["features"]].values.tolist() if model.controls else None,
                        include_text=False, include_filename=True, device=dev, **pre)
        loader = TTSDataLoader(ds, batch_size=batch_size, shuffle=False, drop_last=False)

        def to_dev(b, d):
            out = _to_dev(b, d)
            out["filename"] = b[2]["filename"]
            return out
        for b in DevicePrefetcher(loader, to_dev, dev):
            args = {k: b[k] for k in ("speaker_id", "controls", "description_embeddings") if k in b}
            with torch.no_grad():
                _, post, _, _ = model(chars_idx=b["chars_idx"], chars_idx_len=b["chars_idx_len"], teacher_forcing=True,
                                      mel_spectrogram=b["mel_spectrogram"], mel_spectrogram_len=b["mel_spectrogram_len"], **args)
            post = post.cpu()
            model.tacotron2._engine.check_persistent_kernels()     # (the copy above synchronised) never export a poisoned forward
            for mel_out, n, fn in zip(post, b["mel_spectrogram_len"].cpu().tolist(), b["filename"]):
                path = os.path.join(results_dir, f"{fn.replace('/', '_')}.np")
                np.save(path, mel_out[:int(n)].numpy())
                written.append(path + ".npy")
    return written
