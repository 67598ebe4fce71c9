"""Tacotron2: drop-in for the reference's model.tacotron2.Tacotron2 (constructor kwargs model/tacotron2.py:15-34,
forward signature and 4-tuple return :155-166,347, state_dict keys SURVEY.md Appendix A) on the gfx950 HIP engine.

All learnable tensors are views into ONE flat fp32 device buffer (tacotron2_amd.params.ParamStore); they are exposed as
ordinary nn.Parameters under the reference's module paths, so state_dict()/load_state_dict()/optimizers work unchanged.
forward(teacher_forcing=True) is differentiable: autograd receives the hand-written backward through one
torch.autograd.Function.  forward(teacher_forcing=False, max_len_override=N) is the autoregressive path.
There is no CPU fallback: the module can be constructed and (de)serialised anywhere, but forward needs the GPU library.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .._lib import call
from ..engine import Engine, _stream
from ..init import init_parameters
from ..params import ParamStore


class _Node(nn.Module):
    """Bare container used to reproduce the reference's module paths (encoder.convolutions.0.weight, ...)."""


def _attach(root: nn.Module, dotted: str, value, buffer: bool = False):
    import weakref
    from .submodules import CLASSES
    parts = dotted.split(".")
    mod = root
    for i, part in enumerate(parts[:-1]):
        if part not in mod._modules:
            cls = CLASSES.get(".".join(parts[:i + 1]), _Node)
            child = cls()
            child.__dict__["_t2_root"] = weakref.ref(root)      # not a registered sub-module: no reference cycle in state
            mod.add_module(part, child)
        mod = mod._modules[part]
    if buffer:
        mod.register_buffer(parts[-1], value)
    else:
        mod.register_parameter(parts[-1], value)


class _TacotronFn(torch.autograd.Function):
    """Teacher-forced forward / backward of the whole model as one autograd node."""

    @staticmethod
    def forward(ctx, model, batch, *params):
        eng: Engine = model._engine
        masks = batch.get("masks")
        if masks is None:
            B, L = batch["chars_idx"].shape
            masks = eng.make_masks(B, L, batch["mel"].shape[1], model.training, model._seed, model._calls)
        model._calls += 1
        outs, ectx = eng.forward_tf(batch["chars_idx"], batch["chars_len"], batch["mel"], batch["mel_len"],
                                    speaker_id=batch.get("speaker_id"),
                                    description_embeddings=batch.get("description_embeddings"),
                                    training=model.training, masks=masks, save_for_backward=batch["need_grad"],
                                    controls=batch.get("controls"))
        ctx.model, ctx.ectx = model, ectx
        ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, d_mels, d_post, d_gates, _d_align):
        model, ectx = ctx.model, ctx.ectx
        eng: Engine = model._engine
        ps = model.store
        B, T, M = ectx["B"], ectx["T"], model.num_mels
        f = lambda g: g.contiguous() if g is not None else None
        d_post_m = eng.buf("ag.d_post", B, T, M)
        dproj = eng.buf("ag.dproj", T, B, M + 1)
        call("t2_outgrad_pack", f(d_mels), f(d_post), f(d_gates), ectx["mlen32"], d_post_m, dproj, B, T, M, _stream())
        ps.grad.zero_()
        eng.backward_tf(ectx, d_post_m, dproj)
        grads = tuple(ps.G[name] for name in model._param_names)
        return (None, None) + grads


class Tacotron2(nn.Module):
    def __init__(self, num_chars: int, encoded_dim: int, encoder_kernel_size: int, num_mels: int, prenet_dim: int,
                 att_rnn_dim: int, att_dim: int, rnn_hidden_dim: int, postnet_dim: int, dropout: float,
                 speaker_tokens: bool = False, speaker_tokens_dim: Optional[int] = 128, num_speakers: int = 1,
                 controls: bool = False, controls_dim: int = 0, description_embeddings: bool = False,
                 description_embeddings_dim: int = 0, device=None, seed: int = 0):
        super().__init__()
        assert not speaker_tokens or num_speakers is not None, "If speaker tokens are enabled, you must give a num_speakers!"
        assert encoder_kernel_size == 5, "the conv-as-GEMM kernels are specialised for the reference's k=5"
        self.embedding_dim = self.char_embedding_dim = encoded_dim
        self.num_mels, self.att_rnn_dim, self.rnn_hidden_dim = num_mels, att_rnn_dim, rnn_hidden_dim
        self.controls, self.controls_dim = controls, controls_dim
        self.speaker_tokens, self.description_embeddings = speaker_tokens, description_embeddings
        self.encoded_full_dim = encoded_dim + (128 if description_embeddings else 0)
        self.dims = dict(num_chars=num_chars, encoded_dim=encoded_dim, encoder_kernel_size=encoder_kernel_size,
                         num_mels=num_mels, prenet_dim=prenet_dim, att_rnn_dim=att_rnn_dim, att_dim=att_dim,
                         rnn_hidden_dim=rnn_hidden_dim, postnet_dim=postnet_dim, dropout=dropout,
                         speaker_tokens=speaker_tokens, num_speakers=num_speakers,
                         description_embeddings=description_embeddings,
                         description_embeddings_dim=description_embeddings_dim,
                         controls=bool(controls), controls_dim=int(controls_dim) if controls else 0)
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        self._seed, self._calls = seed, 0
        self._install(ParamStore(self.dims, device))
        init_parameters(self.store, seed)

    # ---- parameter plumbing -------------------------------------------------------------------------
    def _install(self, store: ParamStore):
        for name in list(self._modules):
            del self._modules[name]
        self.store = store
        self._engine = Engine(store)
        self._param_names = list(store.P)
        for name, view in store.P.items():
            _attach(self, name, nn.Parameter(view, requires_grad=True))
        for name, view in store.Bf.items():
            _attach(self, name, view, buffer=True)
        for name, val in store.num_batches_tracked.items():
            _attach(self, name, torch.tensor(val, dtype=torch.int64), buffer=True)

    def _apply(self, fn, recurse=True):
        """.to()/.cuda()/.cpu(): move the flat buffers and rebuild the views (the aliasing is the point)."""
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.store.device))
        if probe.device != self.store.device:
            new = ParamStore(self.dims, probe.device)
            new.flat.copy_(self.store.flat)
            new.buf_flat.copy_(self.store.buf_flat)
            new.num_batches_tracked = dict(self.store.num_batches_tracked)
            self._install(new)
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing = self.store.load_state_dict(state_dict, strict=False)
        unexpected = [k for k in state_dict if k not in self.store.P and k not in self.store.Bf
                      and k not in self.store.num_batches_tracked]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:4]} unexpected {unexpected[:4]}")
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False):
        sd = self.store.state_dict(prefix)
        if destination is not None:
            destination.update(sd)
            return destination
        return sd

    # ---- forward ------------------------------------------------------------------------------------
    def init_hidden(self, encoded_len: int, batch_size: int, device):
        """model/tacotron2.py:126-153 (zero states; the engine keeps its own, this is for API parity)."""
        z = lambda *s: torch.zeros(*s, device=device)
        return ((z(batch_size, self.att_rnn_dim), z(batch_size, self.att_rnn_dim)), z(batch_size, self.encoded_full_dim),
                z(batch_size, encoded_len), z(batch_size, encoded_len),
                (z(batch_size, self.rnn_hidden_dim), z(batch_size, self.rnn_hidden_dim)))

    def forward(self, chars_idx: Tensor, chars_idx_len: Tensor, teacher_forcing: bool,
                mel_spectrogram: Optional[Tensor] = None, mel_spectrogram_len: Optional[Tensor] = None,
                speaker_id: Optional[Tensor] = None, controls: Optional[Tensor] = None,
                max_len_override: Optional[int] = None, description_embeddings: Optional[Tensor] = None,
                dropout_masks: Optional[dict] = None):
        if teacher_forcing:
            assert mel_spectrogram is not None, "Ground-truth Mel spectrogram is required for teacher forcing"
            assert mel_spectrogram_len is not None, "Ground-truth Mel spectrogram lengths are required for teacher forcing"
        assert not self.speaker_tokens or speaker_id is not None, "speaker_id tensor required when speaker tokens are active!"
        assert not self.description_embeddings or description_embeddings is not None, \
            "description tensor required when description tokens are active!"
        assert not self.controls or controls is not None, "Controls are enabled, but no control vector was passed to the model!"
        assert self.controls or controls is None, "Controls are disabled, but a control vector was passed to the model!"
        if max_len_override is None and mel_spectrogram is None:
            raise Exception("If Mel spectrogram is not given, max_len_override is required!")
        if not chars_idx.is_cuda:
            raise RuntimeError("Tacotron2.forward needs CUDA/HIP tensors: the product path has no CPU fallback")
        if teacher_forcing:
            T = mel_spectrogram.shape[1] if max_len_override is None else max_len_override
            mel = mel_spectrogram[:, :T].contiguous().float()
            batch = dict(chars_idx=chars_idx.contiguous(), chars_len=chars_idx_len, mel=mel, mel_len=mel_spectrogram_len,
                         speaker_id=speaker_id,
                         description_embeddings=description_embeddings.contiguous().float()
                         if description_embeddings is not None else None, masks=dropout_masks,
                         controls=controls, need_grad=torch.is_grad_enabled())
            named = dict(self.named_parameters())
            params = [named[n] for n in self._param_names]      # store order = order of the returned gradients
            return _TacotronFn.apply(self, batch, *params)
        with torch.no_grad():     # ONE loop over the whole batch (groups of 64 in lock-step inside the engine)
            pm = dropout_masks.get("prenet_drop") if dropout_masks else None
            o = self._engine.infer(chars_idx.contiguous(), chars_idx_len, int(max_len_override), speaker_id=speaker_id,
                                   description_embeddings=description_embeddings.contiguous().float()
                                   if description_embeddings is not None else None,
                                   training=self.training, prenet_masks=pm, seed=self._seed + self._calls, controls=controls)
            self._calls += 1
        return o[:4]

    def inference(self, chars_idx, chars_idx_len, max_len: int = 5000, **kw):
        """Alias of forward(teacher_forcing=False, max_len_override=max_len) (north_star's "inference() surface")."""
        return self.forward(chars_idx, chars_idx_len, False, max_len_override=max_len, **kw)
