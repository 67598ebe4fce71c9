"""End-to-end drivers on the GPU: `main.py train` (synthetic batches and a tiny real WAV manifest through the device
log-mel) then `main.py say` from the written Lightning-layout checkpoint."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOWED = "!'(),.:;? \\-abcdefghijklmnopqrstuvwxyz"


def _cfg(tmp_path, train_csv="none.csv", batch=4, controls=None):
    cfg = {"dataset": {"train": str(train_csv), "val": str(train_csv),
                       "preprocessing": {"allowed_chars": ALLOWED, "expand_abbreviations": True, "end_token": "^",
                                         "silence": 512, "trim": False, "num_mels": 80, "cache": True}},
           "training": {"lr": 1e-3, "batch_size": batch, "weight_decay": 1e-6, "name": "tiny", "precision": "16-mixed",
                        "args": {"max_steps": 6, "val_check_interval": 0.5}},
           "model": {"scheduler_milestones": [0.5, 0.75],
                     "args": {"prenet_dim": 32, "att_rnn_dim": 64, "att_dim": 32, "rnn_hidden_dim": 64, "postnet_dim": 64,
                              "dropout": 0.5, "char_embedding_dim": 64, "encoder_kernel_size": 5}},
           "extensions": {"speaker_tokens": {"active": True, "num_speakers": 4}, "controls": {"active": False}}}
    if controls:
        cfg["extensions"]["controls"] = {"active": True, "features": list(controls)}
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    return p


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_cli_train_synthetic_then_say(tmp_path):
    cfg = _cfg(tmp_path)
    res = tmp_path / "res"
    out = _run(["--config", str(cfg), "--device", "0", "train", "--speech-dir", "unused", "--results-dir", str(res),
                "--synthetic", "--max-steps", "3"])
    assert "training_loss" in out and os.path.exists(res / "final.ckpt")
    ck = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert all(k.startswith("tacotron2.") for k in ck["state_dict"]) and ck["hyper_parameters"]["encoded_dim"] == 64
    npy = tmp_path / "say.npy"
    _run(["--config", str(cfg), "--device", "0", "say", "--checkpoint", str(res / "final.ckpt"), "--text",
          "Hello, Mr. Smith-Jones!", "--out", str(npy), "--random-seed", "3", "--speaker-id", "1"])
    # an untrained model never emits a stop, so `say` runs to its 5000-frame cap like the reference would
    mel = np.load(npy)
    assert mel.ndim == 2 and mel.shape[1] == 80 and np.isfinite(mel).all()
    # the same call with a .wav target runs the Griffin-Lim branch of run/say.py:161-173 (16-bit PCM at the dataset's rate)
    wav = tmp_path / "say.wav"
    _run(["--config", str(cfg), "--device", "0", "say", "--checkpoint", str(res / "final.ckpt"), "--text",
          "Hello, Mr. Smith-Jones!", "--out", str(wav), "--random-seed", "3", "--speaker-id", "1"])
    import wave
    with wave.open(str(wav), "rb") as w:
        assert w.getframerate() == 22050 and w.getsampwidth() == 2 and w.getnframes() == 256 * (mel.shape[0] - 1)
    # --hifi-gan-checkpoint (run/say.py:66-86,153-159): a generator checkpoint in the published layout ({"generator": weight-normed
    # state_dict}, config.json next to it; UNIVERSAL_V1 strides 8*8*2*2 = 256 samples per frame, narrow channels for speed)
    from helpers import write_hifigan_checkpoint
    gck = write_hifigan_checkpoint(str(tmp_path / "hifi"))
    hwav = tmp_path / "say_hifi.wav"
    _run(["--config", str(cfg), "--device", "0", "say", "--checkpoint", str(res / "final.ckpt"), "--text",
          "Hello, Mr. Smith-Jones!", "--out", str(hwav), "--random-seed", "3", "--speaker-id", "1", "--hifi-gan-checkpoint", gck])
    with wave.open(str(hwav), "rb") as w:
        assert w.getframerate() == 22050 and w.getnframes() == 256 * mel.shape[0]


def test_cli_train_on_wav_manifest_resume(tmp_path):
    sr = 22050
    speech = tmp_path / "wavs"
    speech.mkdir()
    rows = ["text|wav|speaker_id"]
    rng = np.random.default_rng(0)
    for i in range(6):
        n = sr // 2 + 997 * i
        x = 0.3 * np.sin(2 * np.pi * (200 + 40 * i) * np.arange(n) / sr) + 0.01 * rng.normal(size=n)
        with wave.open(str(speech / f"u{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        rows.append(f"Utterance number {i}, Dr. Who says hi!|u{i}.wav|{i % 4}")
    csvp = tmp_path / "train.csv"
    csvp.write_text("\n".join(rows) + "\n")
    cfg = _cfg(tmp_path, csvp, batch=3)
    res = tmp_path / "res"
    _run(["--config", str(cfg), "train", "--speech-dir", str(speech), "--results-dir", str(res), "--max-steps", "2"])
    assert os.path.exists(res / "final.ckpt") and len(os.listdir(res / "mel_cache")) >= 3
    # the resumed run takes the item-at-a-time loader (training.loader = "items": TTSDataset.__getitem__ + collate, the reference's
    # own structure, kept for A/B runs; the first run used the default batched device loader) - and finds the first run's cache
    c = json.loads(cfg.read_text()); c["training"]["loader"] = "items"; cfg.write_text(json.dumps(c))
    out = _run(["--config", str(cfg), "train", "--speech-dir", str(speech), "--results-dir", str(res), "--max-steps", "4",
                "--resume-ckpt", str(res / "final.ckpt")])
    assert "step 3/4" in out or "step 4/4" in out
    assert "input pipeline:" not in out                    # (that line is the batched loader's; the first run printed it)
    # train-mel-export (run/train_mel_export.py): teacher-forced post-net mels of train + val manifests, one file per wav
    exp = tmp_path / "export"
    _run(["--config", str(cfg), "train-mel-export", "--speech-dir", str(speech), "--checkpoint", str(res / "final.ckpt"),
          "--results-dir", str(exp)])
    for i in range(6):
        m = np.load(exp / f"u{i}.wav.np.npy")
        n = sr // 2 + 997 * i + 512                        # samples after the configured silence pad
        assert m.shape == (1 + n // 256, 80) and np.isfinite(m).all()


def test_cli_controls_extension_train_then_say(tmp_path):
    """extensions.controls (run/train.py:78-83,176-180; run/say.py:113-118): feature columns of the manifest become the
    per-utterance controls vector in training; `say --controls a,b` feeds one at synthesis time."""
    sr = 22050
    speech = tmp_path / "wavs"
    speech.mkdir()
    rows = ["text|wav|speaker_id|pitch_speaker_norm|rate_speaker_norm"]
    for i in range(4):
        n = sr // 2 + 733 * i
        x = 0.3 * np.sin(2 * np.pi * (180 + 30 * i) * np.arange(n) / sr)
        with wave.open(str(speech / f"c{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        rows.append(f"Controlled utterance {i}.|c{i}.wav|{i % 4}|{0.5 * i - 1.0}|{1.0 - 0.25 * i}")
    csvp = tmp_path / "train.csv"
    csvp.write_text("\n".join(rows) + "\n")
    cfg = _cfg(tmp_path, csvp, batch=2, controls=("pitch_speaker_norm", "rate_speaker_norm"))
    res = tmp_path / "res"
    _run(["--config", str(cfg), "train", "--speech-dir", str(speech), "--results-dir", str(res), "--max-steps", "2"])
    ck = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert ck["hyper_parameters"]["controls"] and ck["hyper_parameters"]["controls_dim"] == 2
    assert ck["state_dict"]["tacotron2.decoder.lstm.weight_ih"].shape == (4 * 64, 64 + 64 + 2)       # reference layout
    assert ck["state_dict"]["tacotron2.decoder.mel_out.weight"].shape == (80, 64 + 64 + 2)
    outs = []
    for vals in ("0.5,-1.0", "-2.0,3.0"):
        npy = tmp_path / f"say_{len(outs)}.npy"
        _run(["--config", str(cfg), "--device", "0", "say", "--checkpoint", str(res / "final.ckpt"), "--text", "Hi there.",
              "--out", str(npy), "--random-seed", "3", "--speaker-id", "1", "--controls", vals])
        outs.append(np.load(npy))
    # (an untrained model stops at once, so the two outputs carry no signal to compare; the numerics of the controls path are
    # pinned by tests/golden/tf_train_ctrl.npz and infer_ctrl.npz in test_gpu_model.py)
    assert all(np.isfinite(o).all() and o.shape[1] == 80 for o in outs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", str(cfg), "--device", "0", "say", "--checkpoint",
                        str(res / "final.ckpt"), "--text", "Hi.", "--out", str(tmp_path / "bad.npy"), "--controls", "1.0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "comma-separated" in (r.stdout + r.stderr)      # wrong number of control values


def _wav_manifest(tmp_path, n=6, sr=22050):
    speech = tmp_path / "wavs"
    speech.mkdir()
    rows = ["text|wav|speaker_id"]
    rng = np.random.default_rng(0)
    for i in range(n):
        k = sr // 2 + 997 * i
        x = 0.3 * np.sin(2 * np.pi * (200 + 40 * i) * np.arange(k) / sr) + 0.01 * rng.normal(size=k)
        with wave.open(str(speech / f"u{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        rows.append(f"Utterance number {i}{', and a few more words' * (i % 3)}, Dr. Who says hi!|u{i}.wav|{i % 4}")
    csvp = tmp_path / "train.csv"
    csvp.write_text("\n".join(rows) + "\n")
    return speech, csvp


def test_cli_train_data_parallel_step_over_rccl_never_blocks_the_host(tmp_path):
    """`main.py train` with the data-parallel step forced on at world size 1 (T2_FORCE_DP=1: process group over RCCL, shape agreed on
    the host over the trainer's gloo group, both gradient buckets, Work.wait) and torch's sync-debug mode set to "error" around
    every Trainer.train_step (T2_SYNC_DEBUG=1): the run only succeeds if no call of the step synchronises the host - no `.item()`,
    no `int()` of a device value, no blocking copy - between two loss printouts (run/train.py's loop)."""
    cfg = _cfg(tmp_path)
    res = tmp_path / "res"
    env = dict(os.environ, T2_FORCE_DP="1", T2_SYNC_DEBUG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29661")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", str(cfg), "--device", "0", "train", "--speech-dir",
                        "unused", "--results-dir", str(res), "--synthetic", "--max-steps", "5"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    losses = [float(l.split("training_loss ")[1].split()[0]) for l in r.stdout.splitlines() if "training_loss" in l]
    assert len(losses) >= 2 and all(np.isfinite(losses))
    ck = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert ck["global_step"] == 5


@pytest.mark.gpu_processes(3)
def test_cli_train_two_ranks_on_a_wav_manifest(tmp_path):
    """`main.py train` as two ranks (torch.distributed.run; gloo, both on cuda:0 - RCCL needs a GPU per rank) on a real WAV
    manifest: utterances sharded by rank (3 each, batch 1: an epoch is 3 steps, the 5 steps cross it), every batch padded to the
    step's global (L, T) by the LOADER THREADS (DevicePrefetcher(negotiate=Trainer.negotiate_collated, limit, cycle)) while the
    previous step runs, train_step(padded=True) on both ranks, the flat-gradient all-reduce, one checkpoint from rank 0."""
    speech, csvp = _wav_manifest(tmp_path)
    cfg = _cfg(tmp_path, csvp, batch=1)
    res = tmp_path / "res"
    env = dict(os.environ, T2_DIST_BACKEND="gloo", T2_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29663", os.path.join(ROOT, "main.py"), "--config", str(cfg), "--device", "0", "train", "--speech-dir",
           str(speech), "--results-dir", str(res), "--max-steps", "5"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    losses = [float(l.split("training_loss ")[1].split()[0]) for l in r.stdout.splitlines() if "training_loss" in l]
    assert len(losses) >= 2 and all(np.isfinite(losses))
    ck = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert ck["global_step"] == 5 and "saved" in r.stdout


@pytest.mark.gpu_processes(3)
def test_bench_two_ranks_rehearsal_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (no WORLD_SIZE): bench.py starts its own two ranks as fresh child
    processes before any GPU call (launch_ranks) and relays rank 0's line.  The data-parallel code path end to end on the GPU
    (shard padding, flat-gradient all-reduce, 1/world scale) with both ranks sharing cuda:0 over gloo (RCCL needs one GPU per
    rank; the 8-GPU run is the driver's)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                     # rank 0's line only reaches stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["allreduce"] == "1 call after the backward"            # the default (the four-rank rehearsal runs the two buckets)
    assert all(np.isfinite(d["loss"]))
    # attribution of a scaling shortfall: one record per rank (its own wall time, GPU time of a step, the all-reduce segment), and
    # the stream-concurrency check of every rank's process
    pr = d["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["ms_per_step"] > 0 and x["gpu_step_ms"] > 0 and x["allreduce_ms"] >= 0 for x in pr)
    assert d["config"]["queue_check"]["ok"] is True and all(x["queue_check_ok"] for x in pr)
    assert "launcher: started 2 ranks" in r.stderr
    # the other all-reduce mode is timed beside the judged line, so that a multi-GPU run can settle the default
    assert d["allreduce_alone"]["bytes"] > 1e6 and d["allreduce_alone"]["ms"] > 0
    om = d["other_allreduce_mode"]
    assert om["allreduce"].startswith("2 buckets") and om["ms_per_step"] > 0


@pytest.mark.gpu_processes(5)
def test_bench_four_ranks_rehearsal_on_one_gpu(tmp_path):
    """`bench.py --gpus 4` under an EXTERNAL launcher (torch.distributed.run: WORLD_SIZE set, bench.py is one rank - the form the
    task statement gives for the driver; the two-rank rehearsal above covers the plain command line), rehearsed with four ranks on cuda:0 over gloo (the GPU box admits six
    processes on its card; the 8-rank run needs the 8-GPU node and is the driver's): global batch = 4 x the per-rank batch,
    every rank padded to the global shape, the two-bucket all-reduce (`--overlap-allreduce`), one JSON line from rank 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29614", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--batch", "3", "--backend", "gloo", "--share-gpu", "--overlap-allreduce"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-12000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 12 and d["config"]["parallelism"] == "dp4"
    assert d["value"] > 0 and d["scaling"] == "weak" and all(np.isfinite(d["loss"]))
    assert d["config"]["sync_batchnorm"] is False and d["config"]["allreduce"] == "2 buckets, tail overlapped with the encoder backward"
    assert "cpu_baseline" not in d and d["decode"] is None                     # N = 1 only


def test_cli_vanilla_ljspeech_stop_config_batch2_ten_steps(tmp_path):
    """BASELINE configs[0]: config/vanilla-ljspeech-stop.json at batch 2 for 10 train steps.  The reference runs this on
    CPU; this package has no CPU product path by design (the oracle must not become one), so the configuration is
    exercised on the GPU: the reference file's own values and STALE schema (`char_embedding_dim`, no
    `extensions.descriptions`, speaker tokens inactive, a `test` manifest key; config/vanilla-ljspeech-stop.json:1-52,
    run/train.py:70,87,118-122) at the full vanilla dims, synthetic LJSpeech-shaped batches, then `say` from the result."""
    cfg = {"dataset": {"train": "data/ljspeech-train-v4.csv", "test": "data/ljspeech-test-v4.csv", "val": "data/ljspeech-val-v4.csv",
                       "preprocessing": {"allowed_chars": ALLOWED, "expand_abbreviations": True, "end_token": "^", "silence": 512,
                                         "trim": False, "num_mels": 80, "cache": False}},
           "training": {"lr": 0.001, "batch_size": 2, "weight_decay": 0.000001, "precision": "16-mixed",
                        "name": "vanilla-ljspeech-stop", "float32_matmul_precision": "high", "stopping_val_loss_threshold": None,
                        "args": {"max_steps": 100000}},
           "model": {"scheduler_milestones": [0.5, 0.75],
                     "args": {"prenet_dim": 256, "att_rnn_dim": 1024, "att_dim": 128, "rnn_hidden_dim": 1024, "postnet_dim": 512,
                              "dropout": 0.5, "char_embedding_dim": 512, "encoder_kernel_size": 5}},
           "extensions": {"speaker_tokens": {"active": False}, "controls": {"active": False}}}
    p = tmp_path / "vanilla-ljspeech-stop.json"
    p.write_text(json.dumps(cfg))
    res = tmp_path / "res"
    out = _run(["--config", str(p), "--device", "0", "train", "--speech-dir", "unused", "--results-dir", str(res),
                "--synthetic", "--max-steps", "10"])
    losses = [float(l.split("training_loss ")[1].split()[0]) for l in out.splitlines() if "training_loss" in l]
    assert len(losses) >= 2 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    ck = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert ck["global_step"] == 10 and ck["hyper_parameters"]["encoded_dim"] == 512 and not ck["hyper_parameters"]["speaker_tokens"]
    assert ck["state_dict"]["tacotron2.decoder.att_rnn.weight_ih"].shape == (4096, 256 + 512)
    assert "tacotron2.speaker_embedding.weight" not in ck["state_dict"]
    assert len(ck["optimizer_states"][0]["state"]) == 55 and ck["lr_schedulers"][0]["last_epoch"] == 10
    npy = tmp_path / "say.npy"
    _run(["--config", str(p), "--device", "0", "say", "--checkpoint", str(res / "final.ckpt"), "--text", "Hello there.",
          "--out", str(npy), "--random-seed", "1"])
    mel = np.load(npy)
    assert mel.ndim == 2 and mel.shape[1] == 80 and mel.shape[0] >= 1 and np.isfinite(mel).all()


def test_cli_finetune_runs_exactly_n_steps_with_frozen_encoder(tmp_path):
    """run/train.py:109-113,229-233,245: --finetune resumes global_step / Adam / scheduler from the checkpoint, raises max_steps
    by --finetune-steps (so exactly that many optimiser steps run), doubles the batch, freezes encoder + speaker embedding."""
    cfg = _cfg(tmp_path)
    c = json.loads(cfg.read_text()); c["training"]["args"]["max_steps"] = 4; cfg.write_text(json.dumps(c))
    res = tmp_path / "res"
    _run(["--config", str(cfg), "--device", "0", "train", "--speech-dir", "unused", "--results-dir", str(res), "--synthetic"])
    ck0 = torch.load(res / "final.ckpt", map_location="cpu", weights_only=True)
    assert ck0["global_step"] == 4 and sorted(ck0["lr_schedulers"][0]["milestones"].elements()) == [2, 3]
    out = _run(["--config", str(cfg), "--device", "0", "train", "--speech-dir", "unused", "--results-dir", str(res), "--synthetic",
                "--finetune", "--finetune-steps", "3", "--resume-ckpt", str(res / "final.ckpt")])
    assert "(3 optimiser steps this run, global_step 7)" in out, out[-1500:]
    assert "lr 1.00e-05" in out          # the restored scheduler state (both milestones passed) sets the rate, as in Lightning
    ck1 = torch.load(res / "finetuned.ckpt", map_location="cpu", weights_only=True)
    assert ck1["global_step"] == 7
    sd0, sd1 = ck0["state_dict"], ck1["state_dict"]
    for k in sd0:
        frozen = k.startswith("tacotron2.encoder.") or k.startswith("tacotron2.speaker_embedding.")
        same = torch.equal(sd0[k], sd1[k])
        if frozen and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert same, k                                     # frozen tensors do not move (BN buffers still track batches)
        if k in ("tacotron2.decoder.lstm.weight_hh", "tacotron2.prenet.0.weight", "tacotron2.postnet.postnet.0.weight"):
            assert not same, k
    # the frozen tensors' Adam moments are untouched as well (torch.optim.Adam skips parameters without a gradient)
    from tacotron2_amd.checkpoint import reference_param_order
    order = reference_param_order(dict(ck1["hyper_parameters"]))
    i = order.index("encoder.convolutions.0.weight")
    assert torch.equal(ck0["optimizer_states"][0]["state"][i]["exp_avg"], ck1["optimizer_states"][0]["state"][i]["exp_avg"])


def test_cli_test_command_writes_one_wav_per_manifest_row(tmp_path):
    """`main.py test` (run/test.py:29-227): the test manifest through the batched decode path, then HiFi-GAN or Griffin-Lim.
    The weights are those of the reference-generated `infer` fixture (small dims, stops after a few frames), so utterances DO
    stop; numbering, lengths (`(gate < 0).argmax` frames x 256 samples) and failures.csv follow the reference."""
    import wave
    from helpers import load_golden, params_from, write_hifigan_checkpoint
    z = load_golden("infer")
    sd = {"tacotron2." + k: v for k, v in params_from(z).items()}
    hp = dict(lr=1e-3, weight_decay=1e-6, num_chars=39, encoded_dim=32, num_mels=16, prenet_dim=16, att_rnn_dim=32, att_dim=16,
              rnn_hidden_dim=32, postnet_dim=32, dropout=0.5)
    ck = tmp_path / "m.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "global_step": 0, "epoch": 0}, ck)
    rows = ["text|wav|speaker_id"] + [f"{t}|none{i}.wav|0" for i, t in enumerate(
        ["Hi there.", "A somewhat longer sentence, Dr. Who!", "ok", "Testing: one; two? three.", "The end"])]
    csvp = tmp_path / "test.csv"
    csvp.write_text("\n".join(rows) + "\n")
    cfg = {"dataset": {"train": "none.csv", "val": "none.csv", "test": str(csvp),
                       "preprocessing": {"allowed_chars": ALLOWED, "expand_abbreviations": True, "end_token": "^", "num_mels": 16}},
           "training": {"lr": 1e-3, "batch_size": 4, "weight_decay": 1e-6, "name": "tiny", "args": {"max_steps": 6}},
           "model": {"scheduler_milestones": [], "args": {"prenet_dim": 16, "att_rnn_dim": 32, "att_dim": 16, "rnn_hidden_dim": 32,
                                                          "postnet_dim": 32, "dropout": 0.5, "char_embedding_dim": 32}},
           "extensions": {"speaker_tokens": {"active": False}, "controls": {"active": False}}}
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps(cfg))
    gck = write_hifigan_checkpoint(str(tmp_path / "hifi"), n_mels=16)
    for name, extra in (("hifi", ["--hifi-gan-checkpoint", gck]), ("gl", [])):
        res = tmp_path / f"res_{name}"
        _run(["--config", str(cfgp), "--device", "0", "test", "--speech-dir", "unused", "--checkpoint", str(ck), "--results-dir",
              str(res), "--batch-size", "3", "--max-len", "40"] + extra)
        fails = set()
        if os.path.exists(res / "failures.csv"):
            fails = {int(l.split("|")[0]) for l in open(res / "failures.csv").read().splitlines()}
        n_ok = 0
        for i in range(1, 6):
            p = res / f"{i}.wav"
            if i in fails and name == "gl":
                assert not os.path.exists(p)
                continue
            with wave.open(str(p), "rb") as w:
                assert w.getframerate() == 22050 and w.getsampwidth() == 2
                if i not in fails:      # HiFi-GAN: 256 samples per frame; Griffin-Lim (centred frames): 256 * (frames - 1)
                    assert w.getnframes() % 256 == 0 and w.getnframes() <= 40 * 256 and (name == "gl" or w.getnframes() > 0)
                    n_ok += 1
        assert n_ok >= 3, (name, fails)       # the fixture's stop logit falls below zero within a few frames


def test_cli_test_correlation_sweeps_the_control_overrides(tmp_path):
    """`main.py test-correlation` (run/test_correlation.py:30-250): the sampled test manifest once per control-vector override,
    one directory per override named `str(override)`, numbering from 1 in each.  Weights: the reference-generated `infer_ctrl`
    fixture (5 controls, small dims, utterances stop after a few frames)."""
    import wave
    from helpers import load_golden, params_from, write_hifigan_checkpoint
    from tacotron2_amd.run.test_correlation import feature_overrides
    z = load_golden("infer_ctrl")
    sd = {"tacotron2." + k: v for k, v in params_from(z).items()}
    hp = dict(lr=1e-3, weight_decay=1e-6, num_chars=39, encoded_dim=32, num_mels=16, prenet_dim=16, att_rnn_dim=32, att_dim=16,
              rnn_hidden_dim=32, postnet_dim=32, dropout=0.5, controls=True, controls_dim=5)
    ck = tmp_path / "m.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "global_step": 0, "epoch": 0}, ck)
    feats = ["f0", "f1", "f2", "f3", "f4"]
    rows = ["text|wav|speaker_id|" + "|".join(feats)]
    for spk in (0, 1):
        rows += [f"{t}|none{spk}{i}.wav|{spk}|0.1|0.2|0.3|0.4|0.5" for i, t in enumerate(
            ["Hi there.", "A somewhat longer sentence, Dr. Who!", "ok", "Testing: one; two? three."])]
    csvp = tmp_path / "test.csv"
    csvp.write_text("\n".join(rows) + "\n")
    cfg = {"dataset": {"train": "none.csv", "val": "none.csv", "test": str(csvp),
                       "preprocessing": {"allowed_chars": ALLOWED, "expand_abbreviations": True, "end_token": "^", "num_mels": 16}},
           "training": {"lr": 1e-3, "batch_size": 4, "weight_decay": 1e-6, "name": "tiny", "args": {"max_steps": 6}},
           "model": {"scheduler_milestones": [], "args": {"prenet_dim": 16, "att_rnn_dim": 32, "att_dim": 16, "rnn_hidden_dim": 32,
                                                          "postnet_dim": 32, "dropout": 0.5, "char_embedding_dim": 32}},
           "extensions": {"speaker_tokens": {"active": False}, "controls": {"active": True, "features": feats}}}
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps(cfg))
    gck = write_hifigan_checkpoint(str(tmp_path / "hifi"), n_mels=16)
    res = tmp_path / "res"
    out = _run(["--config", str(cfgp), "--device", "0", "test-correlation", "--speech-dir", "unused", "--checkpoint", str(ck),
                "--results-dir", str(res), "--samples-per-speaker", "3", "--max-len", "40", "--limit-overrides", "3",
                "--hifi-gan-checkpoint", gck])
    want = [str(o) for o in list(feature_overrides(5))[:3]]
    assert sorted(os.listdir(res)) == sorted(want), os.listdir(res)
    assert f"0 / 50: {want[0]}" in out                                         # the reference's progress line (:133)
    lens = {}
    for w_ in want:
        d = res / w_
        fails = set()
        if os.path.exists(d / "failures.csv"):
            fails = {int(l.split("|")[0]) for l in open(d / "failures.csv").read().splitlines()}
        for i in range(1, 7):                                                     # 3 utterances x 2 speakers, numbered from 1
            with wave.open(str(d / f"{i}.wav"), "rb") as w:
                assert w.getframerate() == 22050
                assert (w.getnframes() == 0) == (i in fails) and w.getnframes() % 256 == 0
                lens.setdefault(w_, []).append(w.getnframes())
        assert not os.path.exists(d / "7.wav")
    assert len(lens) == 3


def test_device_prefetcher_yields_the_loader_batches_in_order():
    """DevicePrefetcher (background thread + copy stream in front of the training loop): same batches, same order, every epoch;
    a failing item surfaces in the consumer; leaving the loop early stops the thread."""
    from tacotron2_amd.datasets.tts_dataset import DevicePrefetcher
    dev = torch.device("cuda:0")
    batches = [{"a": torch.full((4, 1000), float(i)), "n": torch.tensor([i])} for i in range(7)]
    to_dev = lambda b, d: {k: v.to(d) for k, v in b.items()}
    pf = DevicePrefetcher(batches, to_dev, dev, depth=2)
    for _ in range(2):
        got = [(float(b["a"].sum()), int(b["n"]), b["a"].device.type) for b in pf]
        assert got == [(4000.0 * i, i, "cuda") for i in range(7)]

    def bad(b, d):
        raise ValueError("boom")
    with pytest.raises(ValueError):
        next(iter(DevicePrefetcher(batches, bad, dev)))
    it = iter(pf)
    assert int(next(it)["n"]) == 0
    it.close()


def test_cli_say_description_takes_a_precomputed_embedding(tmp_path):
    """`say --description PATH` (run/say.py:93-116 feeds BERT's pooler_output of the description text; the weights are remote, so the
    CLI takes the precomputed vector the way the dataset does, datasets/tts_dataset.py:277-287).  Weights of the reference-generated
    `tf_train_desc` fixture (7 speaker tokens, description dim 24): the embedding must change the output against the zero vector, the
    .pt and .npy forms of one vector must give the same output, raw text is refused with the BERT message."""
    from helpers import SMALL, load_golden, params_from
    z = load_golden("tf_train_desc")
    sd = {"tacotron2." + k: v for k, v in params_from(z).items()}
    sd["tacotron2.decoder.gate.bias"] = torch.full_like(sd["tacotron2.decoder.gate.bias"], 50.0)    # never stops: 5000 frames to compare
    hp = dict(lr=1e-3, weight_decay=1e-6, dropout=0.5, speaker_tokens=True, num_speakers=7, description_embeddings=True,
              description_embeddings_dim=24, **SMALL)
    ck = tmp_path / "desc.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "global_step": 0, "epoch": 0}, ck)
    cfg = {"dataset": {"train": "none.csv", "val": "none.csv",
                       "preprocessing": {"allowed_chars": ALLOWED, "expand_abbreviations": True, "end_token": "^", "num_mels": 16}},
           "training": {"lr": 1e-3, "batch_size": 4, "weight_decay": 1e-6, "name": "tiny", "args": {"max_steps": 6}},
           "model": {"scheduler_milestones": [], "args": {"prenet_dim": 16, "att_rnn_dim": 32, "att_dim": 16, "rnn_hidden_dim": 32,
                                                          "postnet_dim": 32, "dropout": 0.5, "char_embedding_dim": 32}},
           "extensions": {"speaker_tokens": {"active": True, "num_speakers": 7}, "controls": {"active": False},
                          "descriptions": {"bert_embeddings": True, "finetuneable": False}}}
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps(cfg))
    v = torch.randn(24, generator=torch.Generator().manual_seed(5)) * 2
    torch.save(v, tmp_path / "d.pt")
    np.save(tmp_path / "d.npy", v.view(1, 24).numpy())
    outs = {}
    for name, extra in (("zeros", []), ("pt", ["--description", str(tmp_path / "d.pt")]), ("npy", ["--description", str(tmp_path / "d.npy")])):
        npy = tmp_path / f"say_{name}.npy"
        _run(["--config", str(cfgp), "--device", "0", "say", "--checkpoint", str(ck), "--text", "Hello there, how are you?", "--out",
              str(npy), "--random-seed", "3", "--speaker-id", "2"] + extra)
        outs[name] = np.load(npy)
        assert outs[name].ndim == 2 and outs[name].shape[1] == 16 and np.isfinite(outs[name]).all()
    assert outs["pt"].shape == outs["npy"].shape and np.array_equal(outs["pt"], outs["npy"])
    assert outs["pt"].shape == outs["zeros"].shape == (4999, 16)        # ran into max_len 5000; the last frame is dropped (run/say.py:155)
    assert np.abs(outs["pt"][:50] - outs["zeros"][:50]).max() > 1e-3    # the description vector conditions the memory of every frame
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", str(cfgp), "--device", "0", "say", "--checkpoint", str(ck),
                        "--text", "Hi.", "--out", str(tmp_path / "bad.npy"), "--description", "a calm, low voice"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "bert-base-uncased" in (r.stdout + r.stderr)
