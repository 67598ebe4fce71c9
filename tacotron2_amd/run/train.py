"""do_train (run/train.py:21-255) without Lightning: CSV manifests -> TTSDataset -> batches -> fused HIP training step
(forward + loss + backward + global-norm clip 1.0 + Adam + MultiStepLR) -> Lightning-layout checkpoint.
Multi-GPU: launch one process per GPU with torch.distributed.run; utterances are sharded by rank and the flat gradient
buffer is all-reduced over RCCL (tacotron2_amd.trainer.Trainer)."""
from __future__ import annotations

import csv
import datetime
import os
import time
from typing import Optional

import torch
import torch.distributed as dist

from ..model.tts_model import TTSModel
from ..trainer import Trainer
from .common import model_kwargs


def _to_dev(batch, dev):
    data, meta, _ = batch
    out = dict(chars_idx=data["chars_idx"].to(dev), chars_idx_len=meta["chars_idx_len"].to(dev),
               mel_spectrogram=data["mel_spectrogram"].to(dev).float().contiguous(),
               mel_spectrogram_len=meta["mel_spectrogram_len"].to(dev), gate=data["gate"].to(dev).float().contiguous())
    for k in ("speaker_id", "description_embeddings"):
        if k in meta:
            out[k] = meta[k].to(dev)
    if "features" in meta:                      # controls extension: model/tts_model.py:125 feeds metadata["features"]
        out["controls"] = meta["features"].to(dev).float()
    return out


def do_train(dataset_config: dict, training_config: dict, model_config: dict, extensions_config: dict, device: int,
             speech_dir: str, results_dir: Optional[str], resume_ckpt: Optional[str], finetune: bool = False,
             finetune_steps: Optional[int] = None, max_steps_override: Optional[int] = None, synthetic: bool = False):
    import pandas as pd
    from ..datasets.tts_dataset import TTSDataLoader, TTSDataset
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        device = int(os.environ.get("LOCAL_RANK", device))
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    if results_dir is None:
        results_dir = f"results_{training_config['name']} {datetime.datetime.now()}"
    if rank == 0:
        os.makedirs(results_dir, exist_ok=True)
    cfg = dict(dataset=dataset_config, training=training_config, model=model_config, extensions=extensions_config)
    if finetune:
        training_config["args"]["max_steps"] += finetune_steps
        training_config["lr"] /= 10
        training_config["batch_size"] *= 2
    kw = model_kwargs(cfg)
    max_steps = max_steps_override or training_config["args"]["max_steps"]
    model = TTSModel(device=dev, **kw)
    start_step = 0
    tr = Trainer(model.tacotron2.store, lr=kw["lr"], weight_decay=kw["weight_decay"],
                 scheduler_milestones=kw["scheduler_milestones"], max_norm=1.0)
    if resume_ckpt:
        ck = torch.load(resume_ckpt, map_location="cpu", weights_only=True)
        model.load_checkpoint_dict(ck)
        if "t2_optimizer" in ck and not finetune:
            st = ck["t2_optimizer"]
            tr.ps.init_adam()
            tr.ps.exp_avg.copy_(st["exp_avg"]); tr.ps.exp_avg_sq.copy_(st["exp_avg_sq"])
            tr.global_step = start_step = int(ck.get("global_step", 0))
    if finetune:   # run/train.py:229-233: encoder and speaker embedding are frozen
        tr.frozen = {n for n in tr.ps.P if n.startswith("encoder.") or n.startswith("speaker_embedding.")}

    if synthetic:
        from ..synthetic import ljspeech_batch
        def batches():
            i = 0
            while True:
                b = ljspeech_batch(training_config["batch_size"], seed=1234 + i * world + rank,
                                   num_speakers=kw["num_speakers"] if kw["speaker_tokens"] else 0)
                i += 1
                yield {k: v.to(dev) for k, v in b.items()}
    else:
        df = pd.read_csv(dataset_config["train"], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
        df = df.iloc[rank::world].reset_index(drop=True)            # utterance sharding across ranks
        desc = None
        if extensions_config["descriptions"].get("bert_embeddings"):
            desc = [None if (isinstance(x, float)) else x for x in df.description_embedding]
        ds = TTSDataset(filenames=list(df.wav), texts=list(df.text), base_dir=speech_dir,
                        speaker_ids=list(df.speaker_id) if kw["speaker_tokens"] else None,
                        features=df[extensions_config["controls"]["features"]].values.tolist() if kw.get("controls") else None,
                        cache_dir=os.path.join(results_dir, "mel_cache"), description_embeddings=desc, device=dev,
                        **dataset_config["preprocessing"])
        loader = TTSDataLoader(ds, batch_size=training_config["batch_size"], shuffle=True, drop_last=True)
        def batches():
            while True:
                for b in loader:
                    yield _to_dev(b, dev)

    t0, frames = time.time(), 0
    it = batches()
    for step in range(start_step, max_steps):
        batch = next(it)
        loss3, _ = tr.train_step(batch)
        frames += int(batch["mel_spectrogram_len"].sum())
        if rank == 0 and (step % 50 == 0 or step == max_steps - 1):
            l = [float(x) for x in loss3.cpu()]
            dt = time.time() - t0
            print(f"step {step + 1}/{max_steps} training_gate_loss {l[0]:.5f} training_mel_loss {l[1]:.5f} "
                  f"training_mel_post_loss {l[2]:.5f} training_loss {sum(l):.5f} lr {tr.lr_at(step):.2e} "
                  f"{frames * world / max(dt, 1e-9):.0f} mel-frames/s", flush=True)
    if rank == 0:
        ck = model.checkpoint(extra=dict(global_step=tr.global_step,
                                         t2_optimizer=dict(exp_avg=tr.ps.exp_avg.cpu(), exp_avg_sq=tr.ps.exp_avg_sq.cpu())
                                         if tr.ps.exp_avg is not None else {}))
        path = os.path.join(results_dir, "finetuned.ckpt" if finetune else "final.ckpt")
        torch.save(ck, path)
        print(f"saved {path}")
    if world > 1:
        dist.barrier()
    return model
