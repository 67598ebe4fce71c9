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
    from ..datasets.tts_dataset import DeviceBatchLoader, DevicePrefetcher, TTSDataLoader, TTSDataset
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    force_dp = bool(training_config.get("force_collectives", False)) or os.environ.get("T2_FORCE_DP") == "1"
    own_group = (world > 1 or force_dp) and not dist.is_initialized()
    if own_group:
        # T2_DIST_BACKEND=gloo + T2_SHARE_GPU=1: rehearsal of N ranks on ONE card (RCCL needs a GPU per rank); T2_FORCE_DP=1 /
        # training.force_collectives: the collectives of the step also at world size 1 (RCCL on a single-GPU box)
        backend = os.environ.get("T2_DIST_BACKEND", "nccl")
        if os.environ.get("T2_SHARE_GPU") != "1":
            device = int(os.environ.get("LOCAL_RANK", device))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":      # (before any GPU call of this process: the communicator binds to the device eagerly)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        torch.cuda.set_device(device)
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
    # run/train.py:170: torch.set_float32_matmul_precision(training_config["float32_matmul_precision"]) - here the number of bf16
    # partial products the GEMM kernel issues ("highest" when the key is absent; every shipped reference config says "high")
    from ..engine import set_float32_matmul_precision
    set_float32_matmul_precision(training_config.get("float32_matmul_precision", "highest"))
    max_steps = max_steps_override or training_config["args"]["max_steps"]
    model = TTSModel(device=dev, **kw)
    start_step = 0
    # training.sync_batchnorm (not a reference key; Lightning's Trainer(sync_batchnorm=...) name): BatchNorm batch statistics over
    # all ranks' shards, so that N x b utterances give the single-device result on the N*b batch.  Default: per shard.
    tr = Trainer(model.tacotron2.store, lr=kw["lr"], weight_decay=kw["weight_decay"],
                 scheduler_milestones=kw["scheduler_milestones"], max_norm=1.0,
                 sync_bn=bool(training_config.get("sync_batchnorm", False)),
                 overlap_allreduce=bool(training_config.get("overlap_allreduce", False)),
                 force_collectives=force_dp)
    if world > 1 and os.environ.get("T2_SHARE_GPU") == "1":
        # rehearsal of N ranks on ONE card: persistent launches of different processes cannot promise each other co-residency
        tr.engine.dec_chain = "steps"; tr.engine.enc_chain = "steps"
    if resume_ckpt:
        # trainer.fit(ckpt_path=...) (run/train.py:245): weights, global_step, Adam moments and the scheduler state all come
        # back, for plain resumes and for --finetune alike (the fine-tune then runs exactly `finetune_steps` more steps)
        from ..checkpoint import restore_trainer
        ck = torch.load(resume_ckpt, map_location="cpu", weights_only=True)
        model.load_checkpoint_dict(ck)
        restore_trainer(ck, tr)
        start_step = tr.global_step
    if finetune:   # run/train.py:229-233: encoder and speaker embedding are frozen
        tr.frozen = {n for n in tr.ps.P if n.startswith("encoder.") or n.startswith("speaker_embedding.")}

    # Every batch reaches Trainer.train_step already padded to the step's GLOBAL (L, T): the shape is agreed on the host, over the
    # trainer's host-side group, before the batch goes to the device (real data: in the loader thread, a step ahead) - the training
    # loop itself issues no collective whose result the host reads, and reads no device value between two loss printouts.
    if synthetic:
        from ..synthetic import ljspeech_batch
        def batches():
            i = 0
            while True:
                b = ljspeech_batch(training_config["batch_size"], seed=1234 + i * world + rank,
                                   num_speakers=kw["num_speakers"] if kw["speaker_tokens"] else 0)
                i += 1
                b = tr.pad_to(b, *tr.negotiate_shape(b["chars_idx"].shape[1], b["mel_spectrogram"].shape[1]))
                yield {k: v.to(dev) for k, v in b.items()}
    else:
        df = pd.read_csv(dataset_config["train"], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
        bucket_window = int(training_config.get("bucket_window", 0))
        if not bucket_window:
            df = df.iloc[rank::world].reset_index(drop=True)        # utterance sharding across ranks (plain shuffle per rank)
        # (with length buckets the sampler shards: same permutation on every rank, rank r takes the r-th slice of every sorted
        #  super-batch, so that all ranks step through batches of similar length - LengthBucketBatchSampler)
        desc = None
        if extensions_config["descriptions"].get("bert_embeddings"):
            desc = [None if (isinstance(x, float)) else x for x in df.description_embedding]
        ds = TTSDataset(filenames=list(df.wav), texts=list(df.text), base_dir=speech_dir,
                        speaker_ids=list(df.speaker_id) if kw["speaker_tokens"] else None,
                        features=df[extensions_config["controls"]["features"]].values.tolist() if kw.get("controls") else None,
                        cache_dir=os.path.join(results_dir, "mel_cache"), description_embeddings=desc, device=dev,
                        **dataset_config["preprocessing"])
        # training.bucket_window (not a reference key): batches of similar text length, see LengthBucketBatchSampler
        # training.loader (not a reference key): "batched" (default) = DeviceBatchLoader - the batch's WAVs decoded by a thread pool
        # (training.decode_threads, 4), ONE host->device copy, ONE batched log-mel pass into the padded (B, T, 80) tensor, nothing
        # read back; "items" = the reference's item-at-a-time path (TTSDataset.__getitem__ + collate), kept for A/B runs
        batched = training_config.get("loader", "batched") == "batched"
        if batched:
            loader = DeviceBatchLoader(ds, batch_size=training_config["batch_size"], shuffle=True, drop_last=True,
                                       bucket_window=bucket_window, seed=0 if bucket_window else rank, rank=rank, world=world,
                                       decode_threads=int(training_config.get("decode_threads", 4)))
        else:
            loader = TTSDataLoader(ds, batch_size=training_config["batch_size"], shuffle=True, drop_last=True,
                                   bucket_window=bucket_window, seed=0 if bucket_window else rank, rank=rank, world=world)
        # the next batches are prepared by a background thread on its own stream
        prefetch = DevicePrefetcher(loader, (lambda b, d_: b.to_device(d_)) if batched else _to_dev, dev,
                                    depth=int(training_config.get("prefetch_batches", 2)),
                                    negotiate=tr.negotiate_collated if tr.loader_negotiation else None,
                                    limit=max(0, max_steps - start_step),
                                    cycle=True)
        def batches():
            yield from prefetch

    val_loader = None
    if not synthetic and dataset_config.get("val") and os.path.exists(dataset_config["val"]):
        vdf = pd.read_csv(dataset_config["val"], delimiter="|", quoting=csv.QUOTE_NONE, engine="c")
        vdesc = None
        if extensions_config["descriptions"].get("bert_embeddings"):
            vdesc = [None if (isinstance(x, float)) else x for x in vdf.description_embedding]
        vds = TTSDataset(filenames=list(vdf.wav), texts=list(vdf.text), base_dir=speech_dir,
                         speaker_ids=list(vdf.speaker_id) if kw["speaker_tokens"] else None,
                         features=vdf[extensions_config["controls"]["features"]].values.tolist() if kw.get("controls") else None,
                         cache_dir=os.path.join(results_dir, "mel_cache"), description_embeddings=vdesc, device=dev,
                         **dataset_config["preprocessing"])
        val_loader = TTSDataLoader(vds, batch_size=min(64, max(2, len(vds))), shuffle=False, drop_last=False)   # run/train.py:160-168

    def validate():
        """The reference's validation pass (run/train.py:160-168, model/tts_model.py:204-253): teacher-forced forward in eval
        mode over the validation manifest, mean of the per-batch validation losses."""
        model.eval()
        tot, n = 0.0, 0
        for vb in val_loader:
            data, meta, extra = vb
            data = {k: v.to(dev) for k, v in data.items()}
            meta = {k: v.to(dev) for k, v in meta.items()}
            data["mel_spectrogram"] = data["mel_spectrogram"].float().contiguous()
            out = model.validation_step((data, meta, extra), n)
            tot += float(out["loss"]); n += 1
        model.train()
        return tot / max(n, 1)

    # checkpoint / validation cadence: Lightning checkpoints at the end of every epoch and validates every
    # `val_check_interval` (int: steps, float: fraction of an epoch; fine-tuning forces 1.0, run/train.py:112)
    steps_per_epoch = max(1, len(loader)) if not synthetic else int(training_config["args"].get("checkpoint_every_n_steps", 1000))
    vci = 1.0 if finetune else training_config["args"].get("val_check_interval", 1.0)
    val_every = int(vci) if isinstance(vci, int) and not isinstance(vci, bool) else max(1, int(round(float(vci) * steps_per_epoch)))
    ckpt_every = int(training_config["args"].get("checkpoint_every_n_steps", steps_per_epoch))

    def save(path, step):
        from ..checkpoint import lightning_checkpoint, save_atomic
        save_atomic(lightning_checkpoint(model, tr, epoch=step // steps_per_epoch), path)

    def healthy(where, loss=None):
        """Called only where the host synchronises anyway (loss printout, validation, checkpoint): a persistent launch that gave
        up on a wait has poisoned that step's outputs with NaN and its optimiser step was skipped (engine.forward_tf); training
        must stop here instead of going on, let alone saving."""
        tr.engine.check_persistent_kernels()
        model.tacotron2._engine.check_persistent_kernels()      # (the validation pass runs on the module's own engine)
        if loss is not None and not all(x == x and abs(x) != float("inf") for x in loss):
            raise RuntimeError(f"non-finite training loss {loss} at {where}: the optimiser skips such steps, stopping")

    t0 = time.time()
    frames = torch.zeros((), dtype=torch.int64, device=dev)     # counted on the device: no host synchronisation per step
    it = batches()
    steps_done = 0
    log_every = int(training_config["args"].get("log_every_n_steps", 50))
    # (data-parallel without a host-side group - Trainer warned - : nothing was negotiated by the loader thread, train_step agrees
    #  the shape itself on the main thread.  The synthetic generator pads on the main thread already.)
    pre_padded = synthetic or not tr.dp or tr.loader_negotiation
    sync_debug = os.environ.get("T2_SYNC_DEBUG") == "1"      # tests: prove that a training step never blocks the host
    for step in range(start_step, max_steps):
        batch = next(it)
        if sync_debug:
            torch.cuda.set_sync_debug_mode("error")     # (diagnostic: any host synchronisation inside the step raises)
        loss3, _ = tr.train_step(batch, padded=pre_padded)
        if sync_debug:
            torch.cuda.set_sync_debug_mode("default")
        steps_done += 1
        frames += batch["mel_spectrogram_len"].sum()
        if step % log_every == 0 or step == max_steps - 1:
            l = [float(x) for x in loss3.cpu()]                 # the step's only host synchronisation, every log_every steps
            healthy(f"step {step + 1}", l)
            if rank == 0:
                dt = time.time() - t0
                print(f"step {step + 1}/{max_steps} training_gate_loss {l[0]:.5f} training_mel_loss {l[1]:.5f} "
                      f"training_mel_post_loss {l[2]:.5f} training_loss {sum(l):.5f} lr {tr.lr_at(step):.2e} "
                      f"{int(frames) * world / max(dt, 1e-9):.0f} mel-frames/s", flush=True)
        if val_loader is not None and (step + 1) % val_every == 0:
            vl = validate()                       # every rank runs it (keeps the ranks in step); rank 0 reports
            healthy(f"validation after step {step + 1}")
            if rank == 0:
                print(f"step {step + 1}/{max_steps} validation_loss {vl:.5f}", flush=True)
        if rank == 0 and (step + 1) % ckpt_every == 0 and step + 1 < max_steps:
            torch.cuda.synchronize()
            healthy(f"checkpoint after step {step + 1}")
            save(os.path.join(results_dir, "last.ckpt"), step + 1)
    torch.cuda.synchronize()
    if not synthetic:
        ds.flush_cache()                     # (mel-cache entries still on their way to the disk)
        if rank == 0 and hasattr(loader, "decode_s") and loader.batches:
            print(f"input pipeline: {loader.batches} batches decoded and packed in {loader.decode_s:.2f} s of loader-thread time "
                  f"({loader.batches * training_config['batch_size'] / max(loader.decode_s, 1e-9):.0f} utterances/s)", flush=True)
    healthy("the end of training", [float(x) for x in loss3.cpu()] if steps_done else None)
    if rank == 0:
        path = os.path.join(results_dir, "finetuned.ckpt" if finetune else "final.ckpt")
        save(path, max_steps)
        print(f"saved {path} ({steps_done} optimiser steps this run, global_step {tr.global_step})")
    if tr.dp:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    model.steps_done = steps_done
    return model
