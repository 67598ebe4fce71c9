"""do_test_correlation (run/test_correlation.py:30-250): the test manifest synthesised once per control-vector override, to
correlate the requested prosody controls with what comes out.

Kept from the reference: the 51 overrides (:43-49) - a `set` of 5-tuples, every feature swept over round(arange(-1, 1.1, 0.2), 1)
with the others at 0.0, built and iterated exactly as there (the all-zero vector appears once, as `(-0.0, 0.0, 0.0, 0.0, 0.0)`:
the first insertion wins); 200 utterances per speaker drawn with `groupby("speaker_id").sample(200, random_state=9001)`
(:84-91); the force_speaker consistency checks without any filtering (:93-110); the override replaces EVERY utterance's
features (:136-138); batches of 8 (:159); one directory per override named `str(override)` under
`results_<name>_test_correlation <timestamp>` (:132-134,168-169); numbering from 1 per override, `mel_lengths =
(gate < 0).argmax`, HiFi-GAN on the padded batch rows cut at mel_length * 256, an EMPTY wav + a `failures.csv` row for an
utterance that never stopped (:196-213); Griffin-Lim otherwise, with failures logged (:214-250).
Different on purpose: the HiFi-GAN generator comes from --hifi-gan-checkpoint (the reference ignores the option and reads
`web_checkpoints/hifi-gan/UNIVERSAL_V1/g_02500000`, :66-82); the texts go straight from the manifest to the batched decode
path instead of through a dataset that also decodes every test wav."""
from __future__ import annotations

import csv
import datetime
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .test import check_force_speaker, load_test_model, make_vocoders, synthesize_manifest


def feature_overrides(n_features: int = 5):
    """run/test_correlation.py:43-49, verbatim in behaviour: a set (deduplicated, the set's own iteration order)."""
    base = [0.0] * n_features
    out = set()
    for i in range(n_features):
        for j in np.arange(-1, 1.1, 0.2):
            t = base[:]
            t[i] = float(round(j, 1))       # (plain floats: numpy >= 2 would print np.float64(0.6) in the directory names)
            out.add(tuple(t))
    return out


def do_test_correlation(dataset_config: dict, training_config: dict, model_config: dict, extensions_config: dict, device: int,
                        speech_dir: Optional[str], checkpoint: str, hifi_gan_checkpoint: Optional[str] = None,
                        results_dir: Optional[str] = None, samples_per_speaker: int = 200, max_len: int = 5000,
                        limit_overrides: Optional[int] = None, random_seed: Optional[int] = None) -> Dict[str, List[str]]:
    import pandas as pd
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    pre = dataset_config["preprocessing"]
    ctl_cfg = extensions_config.get("controls", {"active": False})
    if not ctl_cfg.get("active"):
        raise Exception("test-correlation sweeps the prosody controls: extensions.controls must be active")
    df = pd.read_csv(dataset_config["test"], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
    df = df.groupby("speaker_id").sample(samples_per_speaker, random_state=9001).reset_index(drop=True)
    check_force_speaker(extensions_config)
    model = load_test_model(dataset_config, training_config, model_config, extensions_config, checkpoint, dev, random_seed)
    overrides = feature_overrides(len(ctl_cfg["features"]))
    base = results_dir or f"results_{training_config['name']}_test_correlation {datetime.datetime.now()}"
    gen, gl, sr = make_vocoders(hifi_gan_checkpoint, pre, dev)
    out: Dict[str, List[str]] = {}
    for k, ov in enumerate(overrides):
        if limit_overrides is not None and k >= limit_overrides:
            break
        print(f"{k} / {len(overrides) - 1}: {ov}")
        feats = [list(ov)] * len(df)
        out[str(ov)] = synthesize_manifest(model, df, pre, speech_dir, os.path.join(base, str(ov)), gen, gl, sr, feats,
                                           batch_size=8, max_len=max_len, random_seed=random_seed, zero_length="correlation")
    return out
