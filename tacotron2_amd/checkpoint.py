"""Lightning-layout checkpoints: what `Trainer.save_checkpoint` writes for the reference's
TTSModel (run/train.py:245-255) and what `trainer.fit(ckpt_path=...)` / `TTSModel.load_from_checkpoint` read back
(run/train.py:245, run/say.py:125-137, model/tts_model.py:46,78-91):

    epoch, global_step, pytorch-lightning_version, state_dict {"tacotron2.<name>"}, loops, callbacks,
    optimizer_states [Adam.state_dict()], lr_schedulers [MultiStepLR.state_dict()], hparams_name, hyper_parameters

The optimizer state is indexed by position in `self.tacotron2.parameters()` (model/tts_model.py:78-81), i.e. the
reference's module registration order (model/tacotron2.py:58-122, model/decoder.py:26-51, model/attention.py:17-32),
not the order of this package's flat buffer: `reference_param_order` restates it (checked against the key order of the
reference-generated fixtures in tests/test_checkpoint.py).  The flat Adam moments are sliced into per-tensor entries on
save and gathered back on load; the prosody-controls columns, stored here as separate blocks, are concatenated to the
reference's shapes (params.CONTROL_SPLITS).
Files are read with torch.load(weights_only=True) only: every value written here is a tensor, number, string, list,
dict or collections.Counter (MultiStepLR's milestones), all of which that loader accepts.
"""
from __future__ import annotations

import os
from collections import Counter
from typing import Dict, List, Optional

import torch

from .params import CONTROL_SPLITS, ParamStore

LIGHTNING_VERSION = "2.5.2"      # the reference's pinned `lightning` (requirements.txt:8)


def reference_param_order(d: dict) -> List[str]:
    """Names of `Tacotron2.parameters()` in the reference's registration order."""
    names: List[str] = []
    if d.get("speaker_tokens"):
        names.append("speaker_embedding.weight")
    names.append("encoder.embedding.weight")
    for i in (0, 4, 8):
        names += [f"encoder.convolutions.{i}.weight", f"encoder.convolutions.{i}.bias",
                  f"encoder.convolutions.{i + 1}.weight", f"encoder.convolutions.{i + 1}.bias"]
    for sfx in ("", "_reverse"):
        names += [f"encoder.lstm.{k}_l0{sfx}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["prenet.0.weight", "prenet.3.weight"]
    if d.get("description_embeddings"):
        names += ["description_embeddings_linear.0.weight", "description_embeddings_linear.0.bias"]
    names.append("att_encoder.weight")
    names += [f"decoder.att_rnn.{k}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["decoder.attention.query_layer.weight", "decoder.attention.v.weight",
              "decoder.attention.location_conv.weight", "decoder.attention.location_dense.weight"]
    names += [f"decoder.lstm.{k}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["decoder.mel_out.weight", "decoder.mel_out.bias", "decoder.gate.weight", "decoder.gate.bias"]
    for li in range(5):
        names += [f"postnet.postnet.{4 * li}.weight", f"postnet.postnet.{4 * li + 1}.weight", f"postnet.postnet.{4 * li + 1}.bias"]
    return names


def _moment_views(ps: ParamStore, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Reference-named, reference-shaped CPU copies of a flat per-parameter buffer (controls columns appended)."""
    table = {}
    for name, shp in ps.shapes.items():
        o = ps.offsets[name]
        n = 1
        for x in shp:
            n *= x
        table[name] = flat[o:o + n].view(shp)
    return {k: v.detach().cpu().clone() for k, v in ps.reference_layout(table).items()}


def optimizer_state(ps: ParamStore, step: int, lr: float, base_lr: float, weight_decay: float, betas=(0.9, 0.999),
                    eps: float = 1e-8) -> dict:
    """torch.optim.Adam(...).state_dict() as the reference's optimizer would hold it after `step` updates."""
    order = reference_param_order(ps.dims)
    state = {}
    if ps.exp_avg is not None and step > 0:
        m, v = _moment_views(ps, ps.exp_avg), _moment_views(ps, ps.exp_avg_sq)
        for i, name in enumerate(order):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[name], "exp_avg_sq": v[name]}
    group = dict(lr=float(lr), betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False,
                 initial_lr=float(base_lr), params=list(range(len(order))))
    return {"state": state, "param_groups": [group]}


def scheduler_state(milestones, base_lr: float, lr: float, step: int) -> dict:
    """MultiStepLR.state_dict() after `step` scheduler steps (interval "step", model/tts_model.py:83-88)."""
    return {"milestones": Counter(int(m) for m in milestones), "gamma": 0.1, "base_lrs": [float(base_lr)],
            "last_epoch": int(step), "_step_count": int(step) + 1, "_get_lr_called_within_step": False,
            "_last_lr": [float(lr)]}


def lightning_checkpoint(model, trainer=None, epoch: int = 0) -> dict:
    """The dict `Trainer.save_checkpoint` writes for this model (model: tacotron2_amd.model.TTSModel)."""
    ps: ParamStore = model.tacotron2.store
    ck = {"epoch": int(epoch), "global_step": int(trainer.global_step) if trainer is not None else 0,
          "pytorch-lightning_version": LIGHTNING_VERSION,
          "state_dict": {k: v.cpu() for k, v in model.tacotron2.state_dict(prefix="tacotron2.").items()},
          "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [],
          "hparams_name": "kwargs", "hyper_parameters": dict(model.hparams)}
    if trainer is not None:
        step = int(trainer.global_step)
        lr = trainer.lr_at(step)
        ck["optimizer_states"] = [optimizer_state(ps, step, lr, trainer.base_lr, trainer.weight_decay)]
        if trainer.milestones:
            ck["lr_schedulers"] = [scheduler_state(trainer.milestones, trainer.base_lr, lr, step)]
    return ck


def save_atomic(ck: dict, path: str) -> None:
    """Write to a temporary file in the same directory, then rename: a crash never leaves a truncated checkpoint."""
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(ck, tmp)
    os.replace(tmp, path)


def restore_trainer(ck: dict, trainer) -> bool:
    """What `trainer.fit(ckpt_path=...)` restores besides the weights: global_step, the Adam moments and the scheduler
    state (its milestones and base lr REPLACE the freshly configured ones, as MultiStepLR.load_state_dict does).
    Accepts Lightning's keys (written by the reference or by `lightning_checkpoint`) and round 1's private
    `t2_optimizer` block.  Returns True if optimizer state was found."""
    ps: ParamStore = trainer.ps
    trainer.global_step = int(ck.get("global_step", 0))
    found = False
    opt = (ck.get("optimizer_states") or [None])[0]
    if opt and opt.get("state"):
        order = reference_param_order(ps.dims)
        ps.init_adam()
        st = opt["state"]
        for i, name in enumerate(order):
            ent = st.get(i, st.get(str(i)))
            if ent is None:                       # a parameter that never received a gradient (frozen): moments stay zero
                continue
            for key, flat in (("exp_avg", ps.exp_avg), ("exp_avg_sq", ps.exp_avg_sq)):
                t = torch.as_tensor(ent[key]).to(torch.float32)
                o = ps.offsets[name]
                if name in CONTROL_SPLITS and (name + "#controls") in ps.offsets:
                    k0 = ps.shapes[name][1]
                    oc, nc = ps.offsets[name + "#controls"], t[:, k0:].numel()
                    flat[oc:oc + nc].copy_(t[:, k0:].reshape(-1))
                    t = t[:, :k0]
                flat[o:o + t.numel()].copy_(t.reshape(-1))
        found = True
    elif "t2_optimizer" in ck and ck["t2_optimizer"]:
        ps.init_adam()
        ps.exp_avg.copy_(ck["t2_optimizer"]["exp_avg"]); ps.exp_avg_sq.copy_(ck["t2_optimizer"]["exp_avg_sq"])
        found = True
    sch = (ck.get("lr_schedulers") or [None])[0]
    if sch:
        ms = sorted(int(m) for m, c in dict(sch["milestones"]).items() for _ in range(int(c)))
        # MultiStepLR counts in ITS epochs.  Normally last_epoch == global_step.  A checkpoint the reference wrote after a resume
        # from a scheduler-less checkpoint (the case below) carries a scheduler that was created at that resume: last_epoch counts
        # from there, its milestones are relative to it, and its base_lrs hold the lr that was configured then - while the lr it
        # actually steps is the optimizer's restored one (MultiStepLR multiplies param_groups[i]["lr"], it does not re-derive it from
        # base_lrs).  Both are brought back to this trainer's absolute step axis: milestones shifted by global_step - last_epoch,
        # the lr level from _last_lr and the milestones already passed.
        last = int(sch.get("last_epoch", trainer.global_step))
        off = trainer.global_step - last
        trainer.milestones = [m + off for m in ms]
        gamma = float(sch.get("gamma", 0.1))
        if off != 0 and sch.get("_last_lr"):
            passed = sum(1 for m in ms if m <= last)
            trainer.base_lr = float(sch["_last_lr"][0]) / (gamma ** passed)
        else:
            trainer.base_lr = float(sch["base_lrs"][0])
    elif opt and opt.get("param_groups"):
        # no scheduler in the checkpoint (it was written with scheduler_milestones = []): Lightning's optimizer.load_state_dict
        # still restores param_groups[0]["lr"], which replaces the freshly configured lr (and the fine-tune's lr / 10) in the
        # reference - and nothing else: restore_lr_schedulers has no state to load, so a MultiStepLR built from the CURRENT config
        # (model/tts_model.py:84-89) stays alive with its own milestones, counted from last_epoch = 0 at the resume point.  The
        # configured milestones are therefore kept, shifted by the restored global_step.
        pg = opt["param_groups"][0]
        trainer.base_lr = float(pg.get("initial_lr", pg["lr"]))
        trainer.milestones = sorted(int(m) + trainer.global_step for m in trainer.milestones)
    return found
