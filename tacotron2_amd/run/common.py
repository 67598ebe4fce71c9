"""Config handling shared by train/say: the reference's four-section JSON (main.py:95-99) with the staleness rules of
SURVEY.md section 5: `char_embedding_dim` is an alias of `encoded_dim`; missing `extensions.*` sections are inactive."""
from __future__ import annotations

import json


def load_config(path: str) -> dict:
    with open(path) as f:
        cfg = json.load(f)
    cfg.setdefault("extensions", {})
    ext = cfg["extensions"]
    ext.setdefault("speaker_tokens", {"active": False})
    ext.setdefault("controls", {"active": False})
    ext.setdefault("descriptions", {"bert_embeddings": False, "finetuneable": False})
    return cfg


def model_kwargs(cfg: dict) -> dict:
    """kwargs for TTSModel from (dataset, training, model, extensions), as run/train.py:176-227 derives them."""
    ds, tr, md, ext = cfg["dataset"], cfg["training"], cfg["model"], cfg["extensions"]
    pre = ds["preprocessing"]
    args = dict(md.get("args", {}))
    if "char_embedding_dim" in args:
        args["encoded_dim"] = args.pop("char_embedding_dim")
    ctl = bool(ext["controls"].get("active"))          # run/train.py:176-180: one control per listed feature column
    args.update(controls=ctl, controls_dim=len(ext["controls"].get("features", [])) if ctl else 0)
    spk = ext["speaker_tokens"].get("active", False)
    max_steps = tr.get("args", {}).get("max_steps", 100000)
    return dict(lr=tr["lr"], weight_decay=tr["weight_decay"],
                num_chars=len(pre["allowed_chars"]) + (pre.get("end_token") is not None),
                num_mels=pre.get("num_mels", 80), speaker_tokens=spk,
                num_speakers=ext["speaker_tokens"].get("num_speakers", 1) if spk else 1,
                scheduler_milestones=[int(x * max_steps) for x in md.get("scheduler_milestones", [])], **args)
