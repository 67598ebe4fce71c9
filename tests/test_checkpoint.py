"""Lightning-layout checkpoint exchange (run/train.py:245-255, run/say.py:125-137,
model/tts_model.py:46,78-91) - host logic, runs without a GPU.

The optimizer state is positional (index in `Tacotron2.parameters()`), so the reference's parameter ORDER is pinned
against the key order of the reference-generated fixtures; the state itself is exchanged with real torch.optim.Adam /
MultiStepLR objects built the way model/tts_model.py:78-91 builds them."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import SMALL, load_golden


def _dims(**extra):
    d = dict(SMALL, dropout=0.5, speaker_tokens=False, num_speakers=1, description_embeddings=False,
             description_embeddings_dim=0, controls=False, controls_dim=0)
    d.update(extra)
    return d


@pytest.mark.parametrize("name,extra", [("tf_train", {}),
                                        ("tf_train_desc", dict(speaker_tokens=True, num_speakers=7, description_embeddings=True,
                                                               description_embeddings_dim=24)),
                                        ("tf_train_ctrl", dict(controls=True, controls_dim=5))])
def test_parameter_order_matches_reference_fixture(name, extra):
    from tacotron2_amd.checkpoint import reference_param_order
    z = load_golden(name)
    is_param = lambda k: not any(k.endswith(s) for s in ("running_mean", "running_var", "num_batches_tracked"))
    ref_order = [k[2:] for k in z if k.startswith("p.") and is_param(k)]          # npz keeps the reference's state_dict order
    assert reference_param_order(_dims(**extra)) == ref_order


def _reference_like_optimizer(d, steps, lr, wd, milestones, seed=0):
    """torch Adam + MultiStepLR over tensors with the reference's names / shapes / order, stepped `steps` times."""
    from oracle import tacotron2_ref as R
    from tacotron2_amd.checkpoint import reference_param_order
    shapes = R.param_shapes(R.default_dims(**d))
    g = torch.Generator().manual_seed(seed)
    order = reference_param_order(d)
    params = [torch.nn.Parameter(torch.randn(shapes[n], generator=g) * 0.1) for n in order]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=milestones, gamma=0.1)
    for _ in range(steps):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g) * 0.01
        opt.step(); sch.step()
    return order, params, opt, sch


@pytest.mark.parametrize("extra", [{}, dict(controls=True, controls_dim=5, speaker_tokens=True, num_speakers=3)])
def test_resume_from_reference_written_checkpoint_restores_adam_and_scheduler(extra, tmp_path):
    """A checkpoint dict laid out as Lightning writes it for the reference (built here from real torch optimizer objects):
    weights, global_step, Adam moments and the scheduler state must come back into the flat buffers."""
    from tacotron2_amd.checkpoint import restore_trainer
    from tacotron2_amd.model.tts_model import TTSModel
    from tacotron2_amd.trainer import Trainer
    d = _dims(**extra)
    order, params, opt, sch = _reference_like_optimizer(d, steps=5, lr=1e-3, wd=1e-6, milestones=[3, 8])
    ck = {"epoch": 0, "global_step": 5, "pytorch-lightning_version": "2.5.2",
          "state_dict": {"tacotron2." + n: p.detach().clone() for n, p in zip(order, params)},
          "loops": {}, "callbacks": {}, "optimizer_states": [opt.state_dict()], "lr_schedulers": [sch.state_dict()],
          "hparams_name": "kwargs", "hyper_parameters": {}}
    path = str(tmp_path / "ref.ckpt")
    torch.save(ck, path)
    ck2 = torch.load(path, map_location="cpu", weights_only=True)                # the only loader the product uses
    model = TTSModel(lr=5e-4, weight_decay=1e-6, scheduler_milestones=[100, 200], device="cpu", **d)
    model.load_checkpoint_dict(ck2, strict=False)                                # BN buffers are absent from this dict
    tr = Trainer(model.tacotron2.store, lr=5e-4, weight_decay=1e-6, scheduler_milestones=[100, 200])
    assert restore_trainer(ck2, tr)
    ps = tr.ps
    assert tr.global_step == 5 and tr.milestones == [3, 8] and tr.base_lr == pytest.approx(1e-3)
    assert tr.lr_at(5) == pytest.approx(sch.get_last_lr()[0])
    table_m = ps.reference_layout({n: ps.exp_avg[ps.offsets[n]:ps.offsets[n] + ps.P[n].numel()].view(ps.shapes[n]) for n in ps.P})
    table_v = ps.reference_layout({n: ps.exp_avg_sq[ps.offsets[n]:ps.offsets[n] + ps.P[n].numel()].view(ps.shapes[n]) for n in ps.P})
    sd = model.tacotron2.state_dict()
    for i, n in enumerate(order):
        st = opt.state[params[i]]
        assert torch.equal(table_m[n], st["exp_avg"]) and torch.equal(table_v[n], st["exp_avg_sq"]), n
        assert torch.equal(sd[n], params[i].detach()), n


def test_written_checkpoint_loads_into_torch_optimizer_and_round_trips(tmp_path):
    """The reverse direction: a checkpoint written here carries every key Lightning's restore path reads, its optimizer and
    scheduler entries load into real torch.optim.Adam / MultiStepLR objects over reference-shaped parameters, and a second
    trainer restored from the file continues with identical moments, step and learning rate."""
    from tacotron2_amd.checkpoint import lightning_checkpoint, reference_param_order, restore_trainer, save_atomic
    from tacotron2_amd.model.tts_model import TTSModel
    from tacotron2_amd.trainer import Trainer
    d = _dims(controls=True, controls_dim=5)
    model = TTSModel(lr=1e-3, weight_decay=1e-6, scheduler_milestones=[4, 9], device="cpu", **d)
    tr = Trainer(model.tacotron2.store, lr=1e-3, weight_decay=1e-6, scheduler_milestones=[4, 9])
    ps = tr.ps
    ps.init_adam()
    g = torch.Generator().manual_seed(1)
    ps.exp_avg.copy_(torch.randn(ps.numel, generator=g)); ps.exp_avg_sq.copy_(torch.rand(ps.numel, generator=g))
    tr.global_step = 6
    path = str(tmp_path / "ours.ckpt")
    save_atomic(lightning_checkpoint(model, tr, epoch=2), path)
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    ck = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks", "optimizer_states",
                "lr_schedulers", "hparams_name", "hyper_parameters"):
        assert key in ck, key
    assert ck["global_step"] == 6 and ck["epoch"] == 2 and ck["hyper_parameters"]["controls_dim"] == 5
    # reference side: model/tts_model.py:78-91 objects accept the entries
    order = reference_param_order(d)
    sd = {k[len("tacotron2."):]: v for k, v in ck["state_dict"].items()}
    params = [torch.nn.Parameter(sd[n].clone()) for n in order]
    assert params[order.index("decoder.lstm.weight_ih")].shape[1] == SMALL["att_rnn_dim"] + SMALL["encoded_dim"] + 5
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[4, 9], gamma=0.1)
    import copy
    opt.load_state_dict(copy.deepcopy(ck["optimizer_states"][0]))                 # (load_state_dict aliases the given tensors)
    sch.load_state_dict(copy.deepcopy(ck["lr_schedulers"][0]))
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-4) and sch.last_epoch == 6
    assert float(opt.state[params[0]]["step"]) == 6.0
    for p in params:
        p.grad = torch.zeros_like(p)
    opt.step(); sch.step()                                                        # and they keep working
    # our side: identical continuation state
    model2 = TTSModel(lr=1e-3, weight_decay=1e-6, scheduler_milestones=[4, 9], device="cpu", **d)
    model2.load_checkpoint_dict(ck)
    tr2 = Trainer(model2.tacotron2.store, lr=1e-3, weight_decay=1e-6, scheduler_milestones=[4, 9])
    assert restore_trainer(ck, tr2)
    assert tr2.global_step == 6 and torch.equal(tr2.ps.flat, ps.flat)
    # alignment padding between tensors is not part of any tensor: compare the tensors' slices
    for n in ps.P:
        o, k = ps.offsets[n], ps.P[n].numel()
        assert torch.equal(tr2.ps.exp_avg[o:o + k], ps.exp_avg[o:o + k]) and torch.equal(tr2.ps.exp_avg_sq[o:o + k], ps.exp_avg_sq[o:o + k])


def test_checkpoint_without_scheduler_restores_the_optimizer_lr(tmp_path):
    """scheduler_milestones = []: no lr_schedulers entry is written (model/tts_model.py:83-90), and on resume Lightning's
    optimizer.load_state_dict brings back param_groups[0]["lr"] - it replaces the freshly configured lr, including the
    fine-tune's lr / 10 (run/train.py:110,245).  Same here.  Nothing else of the scheduler is restored (there is no state to load),
    so the MultiStepLR the CURRENT config builds stays alive and counts its milestones from the resume point
    (model/tts_model.py:84-89, last_epoch = 0): milestone 5 configured, resumed at global_step 3 -> decay from step 8 on."""
    from tacotron2_amd.checkpoint import lightning_checkpoint, restore_trainer, save_atomic
    from tacotron2_amd.model.tts_model import TTSModel
    from tacotron2_amd.trainer import Trainer
    d = _dims()
    model = TTSModel(lr=2e-3, weight_decay=1e-6, scheduler_milestones=[], device="cpu", **d)
    tr = Trainer(model.tacotron2.store, lr=2e-3, weight_decay=1e-6, scheduler_milestones=[])
    tr.ps.init_adam(); tr.global_step = 3
    path = str(tmp_path / "nosched.ckpt")
    save_atomic(lightning_checkpoint(model, tr, epoch=0), path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["lr_schedulers"] == [] and ck["optimizer_states"][0]["param_groups"][0]["lr"] == pytest.approx(2e-3)
    tr2 = Trainer(model.tacotron2.store, lr=2e-4, weight_decay=1e-6, scheduler_milestones=[5])     # fine-tune style: lr / 10
    assert restore_trainer(ck, tr2)
    assert tr2.global_step == 3 and tr2.base_lr == pytest.approx(2e-3) and tr2.milestones == [8]
    assert tr2.lr_at(7) == pytest.approx(2e-3) and tr2.lr_at(8) == pytest.approx(2e-4) and tr2.lr_at(100) == pytest.approx(2e-4)
    tr3 = Trainer(model.tacotron2.store, lr=2e-4, weight_decay=1e-6, scheduler_milestones=[])      # no scheduler configured either
    assert restore_trainer(ck, tr3) and tr3.milestones == [] and tr3.lr_at(100) == pytest.approx(2e-3)


def test_scheduler_whose_epochs_do_not_start_at_step_zero_is_mapped_onto_the_step_axis():
    """The mirror case: a checkpoint written by the reference AFTER such a scheduler-less resume.  Its MultiStepLR was
    created at the resume (last_epoch counts from there, milestones relative to it, base_lrs = the lr configured then) and steps the
    optimizer's RESTORED lr.  Example: resumed at global_step 7 with restored lr 1e-4 under a config of lr 1e-3 / milestones [5, 8];
    saved 3 steps later: global_step 10, last_epoch 3, no milestone passed, _last_lr [1e-4].  Decays are due at steps 12 and 15, from 1e-4."""
    from collections import Counter
    from tacotron2_amd.checkpoint import restore_trainer
    from tacotron2_amd.model.tts_model import TTSModel
    from tacotron2_amd.trainer import Trainer
    model = TTSModel(lr=1e-3, weight_decay=1e-6, scheduler_milestones=[5, 8], device="cpu", **_dims())
    tr = Trainer(model.tacotron2.store, lr=1e-3, weight_decay=1e-6, scheduler_milestones=[5, 8])
    ck = {"global_step": 10, "optimizer_states": [{"state": {}, "param_groups": [{"lr": 1e-4, "initial_lr": 1e-3}]}],
          "lr_schedulers": [{"milestones": Counter({5: 1, 8: 1}), "gamma": 0.1, "base_lrs": [1e-3], "last_epoch": 3, "_last_lr": [1e-4],
                             "_step_count": 4}]}
    restore_trainer(ck, tr)
    assert tr.global_step == 10 and tr.milestones == [12, 15] and tr.base_lr == pytest.approx(1e-4)
    assert tr.lr_at(11) == pytest.approx(1e-4) and tr.lr_at(12) == pytest.approx(1e-5) and tr.lr_at(15) == pytest.approx(1e-6)
    # ... and one milestone already passed since that resume: last_epoch 6 of [5, 8], _last_lr 1e-5 -> level 1e-4 before it
    ck["global_step"], ck["lr_schedulers"][0]["last_epoch"], ck["lr_schedulers"][0]["_last_lr"] = 13, 6, [1e-5]
    restore_trainer(ck, tr)
    assert tr.milestones == [12, 15] and tr.base_lr == pytest.approx(1e-4) and tr.lr_at(13) == pytest.approx(1e-5)
    # the ordinary case is untouched: last_epoch == global_step -> absolute milestones, base_lrs
    ck["global_step"], ck["lr_schedulers"][0]["last_epoch"], ck["lr_schedulers"][0]["_last_lr"] = 6, 6, [1e-4]
    restore_trainer(ck, tr)
    assert tr.milestones == [5, 8] and tr.base_lr == pytest.approx(1e-3) and tr.lr_at(6) == pytest.approx(1e-4)


def test_trainable_ranges_exclude_frozen_tensors():
    from tacotron2_amd.model.tts_model import TTSModel
    from tacotron2_amd.trainer import Trainer
    d = _dims(speaker_tokens=True, num_speakers=3)
    model = TTSModel(lr=1e-3, weight_decay=0.0, device="cpu", **d)
    tr = Trainer(model.tacotron2.store, lr=1e-3, weight_decay=0.0)
    ps = tr.ps
    assert tr.trainable_ranges() == [(0, ps.numel)]
    tr.frozen = {n for n in ps.P if n.startswith("encoder.") or n.startswith("speaker_embedding.")}    # run/train.py:229-233
    rs = tr.trainable_ranges()
    covered = torch.zeros(ps.numel, dtype=torch.bool)
    for a, b in rs:
        assert 0 <= a < b <= ps.numel
        covered[a:b] = True
    for n in ps.P:
        o, k = ps.offsets[n], ps.P[n].numel()
        assert bool(covered[o:o + k].all()) != (n in tr.frozen), n
        assert bool(covered[o:o + k].any()) != (n in tr.frozen), n
    assert len(rs) == 1      # the frozen tensors are one prefix of the flat buffer: the optimizer stays ONE launch
