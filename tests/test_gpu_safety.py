"""The persistent LSTM launches (t2_lstm_seq_fwd_persist: decoder LSTM per chunk, encoder BiLSTM) are kernels whose workgroups wait for each other; every wait
is bounded, and a wait that gives up must stop the training run instead of feeding garbage decoder states into the
optimiser.  These tests force the timeout path (debug bound < 0: every wait counts as timed out) and check the chain of
consequences: sticky device flag -> outputs and loss NaN (t2_guard_poison) -> optimiser step skipped (t2_adam_step with a
non-finite gradient norm) -> the host raises where it reads the loss (Engine.check_persistent_kernels, run/train.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_trainer(dev):
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.synthetic import ljspeech_batch
    from tacotron2_amd.trainer import Trainer
    dims = dict(num_chars=39, encoded_dim=64, encoder_kernel_size=5, num_mels=80, prenet_dim=32, att_rnn_dim=64, att_dim=32,
                rnn_hidden_dim=64, postnet_dim=64, dropout=0.5, speaker_tokens=True, num_speakers=4,
                description_embeddings=False, description_embeddings_dim=0)
    ps = ParamStore(dims, dev); init_parameters(ps, 0)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
    tr.engine.chunk = 16
    batch = {k: v.to(dev) for k, v in ljspeech_batch(4, seed=5, num_speakers=4, fixed_shape=(21, 40)).items()}
    return tr, ps, batch


def test_persistent_launch_residency_is_checked():
    """t2_lstm_persist_resident asks the runtime (compute units x occupancy) instead of assuming 256 free CUs; on a whole
    MI355X the product's launch (H = 1024: 256 workgroups, 72 KB of LDS each) fits, an absurd one does not and reports
    T2_ERR_RESIDENCY so that the engine can fall back to per-step launches."""
    from tacotron2_amd import _lib
    torch.zeros(1, device="cuda:0")
    assert _lib.call_value("t2_lstm_persist_resident", 1024, 1024, 32) == 0
    assert _lib.call_value("t2_lstm_persist_resident", 1024, 1024, 8) == 0
    assert _lib.call_value("t2_lstm_persist_resident", 4 * 4096, 1024, 32) == 3        # 4096 workgroups of 72 KB LDS
    assert b"co-resident" in _lib.lib().t2_last_error()
    # two cells of H = 1024 (an encoder with encoded_dim = 2048) would be 512 workgroups: an answer (use step launches), not an error
    assert _lib.call_value("t2_lstm_persist_resident_n", 1024, 1024, 32, 2) == 3
    assert _lib.call_value("t2_lstm_persist_resident_n", 256, 256, 32, 2) == 0          # the vanilla encoder: 2 x 64 workgroups


def test_engine_falls_back_to_step_launches_without_residency():
    from tacotron2_amd.trainer import Trainer  # noqa: F401
    dev = torch.device("cuda:0")
    tr, ps, batch = _small_trainer(dev)
    loss_p, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    assert tr.engine._persist_sync is not None                     # the persistent path ran
    tr2, ps2, _ = _small_trainer(dev)
    tr2.engine._persist_ok = {(64, True, 1): False, (32, True, 2): False}     # as if t2_lstm_persist_resident had said no (decoder
                                                                                # LSTM H = 64; encoder BiLSTM 2 x H = 32)
    loss_s, _ = tr2.train_step(batch)
    torch.cuda.synchronize()
    assert tr2.engine._persist_sync is None                        # per-step launches on the side stream instead
    assert torch.allclose(loss_p, loss_s, rtol=1e-6, atol=1e-9)
    # same gradients (compared before Adam, whose first update is ~lr * sign(g) and amplifies rounding noise of tiny elements)
    assert float((ps.grad - ps2.grad).abs().max()) < 1e-5 * float(ps.grad.abs().max())


def test_timed_out_persistent_launch_poisons_the_step_and_raises():
    from tacotron2_amd import _lib
    dev = torch.device("cuda:0")
    tr, ps, batch = _small_trainer(dev)
    before = ps.flat.clone()
    old = _lib.call_value("t2_debug_persist_spin_limit", -1)
    try:
        loss3, outs = tr.train_step(batch)
        torch.cuda.synchronize()
    finally:
        _lib.call_value("t2_debug_persist_spin_limit", old)
    assert bool(torch.isnan(loss3).all()), loss3                   # the step announces itself ...
    assert bool(torch.isnan(outs[0]).any())
    assert torch.equal(ps.flat, before)                            # ... and has not touched the weights
    assert tr.engine.ps.exp_avg is None or float(tr.engine.ps.exp_avg.abs().max()) == 0.0
    with pytest.raises(_lib.T2Error, match="timed out"):
        tr.engine.check_persistent_kernels()
    # the check cleared the sticky flag: the next step is a normal one
    loss3, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    tr.engine.check_persistent_kernels()
    assert bool(torch.isfinite(loss3).all()) and not torch.equal(ps.flat, before)


def test_cli_training_stops_on_a_timed_out_persistent_launch(tmp_path):
    """`main.py train` must fail loudly - no final.ckpt - when a persistent launch timed out."""
    from test_gpu_cli import _cfg
    cfg = _cfg(tmp_path)
    res = tmp_path / "res"
    env = dict(os.environ, T2_PERSIST_SPIN_LIMIT="-1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", str(cfg), "--device", "0", "train",
                        "--speech-dir", "unused", "--results-dir", str(res), "--synthetic", "--max-steps", "3"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0, r.stdout[-1500:]
    assert "timed out" in (r.stdout + r.stderr)
    assert not os.path.exists(res / "final.ckpt") and not os.path.exists(res / "last.ckpt")
