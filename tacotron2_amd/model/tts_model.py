"""TTSModel: the reference's training shell (model/tts_model.py) without Lightning: same constructor kwargs, the
`.tacotron2` attribute (state_dict keys `tacotron2.<...>`), forward() delegation (:93-115), the 3-term loss (:197-201),
Adam + MultiStepLR (:78-91).  run/train.py drives the fused HIP step (tacotron2_amd.trainer.Trainer); training_step /
validation_step keep the reference's batch format for external loops."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor, nn

from .._lib import call
from .tacotron2 import Tacotron2


class _LossTermsFn(torch.autograd.Function):
    """(gate BCE-with-logits, mel MSE, post MSE) of model/tts_model.py:197-199 - plain means over the padded tensors - as ONE
    device kernel (t2_loss_terms); with autograd on, the same launch also writes the three dense gradients, which backward scales
    by the upstream gradient of each term."""

    @staticmethod
    def forward(ctx, mel, post, gate, mel_tgt, gate_tgt, mel_len):
        B, T, M = mel.shape
        need = any(ctx.needs_input_grad[:3])
        mel, post, gate = mel.contiguous().float(), post.contiguous().float(), gate.contiguous().float()
        loss3 = torch.empty(3, dtype=torch.float64, device=mel.device)
        grads = [torch.empty_like(t) if need else None for t in (mel, post, gate)]
        call("t2_loss_terms", mel, post, gate, mel_tgt.contiguous().float(), gate_tgt.contiguous().float(),
             mel_len.to(torch.int32), B, T, M, loss3, grads[0], grads[1], grads[2], 1.0, torch.cuda.current_stream().cuda_stream)
        ctx.grads = grads
        return loss3.float()

    @staticmethod
    def backward(ctx, g3):
        d_mel, d_post, d_gate = ctx.grads
        return (d_mel * g3[1] if d_mel is not None else None, d_post * g3[2] if d_post is not None else None,
                d_gate * g3[0] if d_gate is not None else None, None, None, None)


class TTSModel(nn.Module):
    def __init__(self, lr: float, weight_decay: float, num_chars: int, encoded_dim: int = 512, encoder_kernel_size: int = 5,
                 num_mels: int = 80, prenet_dim: int = 256, att_rnn_dim: int = 1024, att_dim: int = 128,
                 rnn_hidden_dim: int = 1024, postnet_dim: int = 512, dropout: float = 0.5,
                 scheduler_milestones: List[int] = (), speaker_tokens: bool = False, num_speakers: int = 1,
                 controls: bool = False, controls_dim: int = 0, max_len_override: Optional[int] = None,
                 description_embeddings: bool = False, description_embeddings_dim: int = 0,
                 char_embedding_dim: Optional[int] = None, device=None):
        super().__init__()
        if char_embedding_dim is not None:     # stale configs name encoded_dim `char_embedding_dim`
            encoded_dim = char_embedding_dim
        self.hparams = dict(lr=lr, weight_decay=weight_decay, num_chars=num_chars, encoded_dim=encoded_dim,
                            encoder_kernel_size=encoder_kernel_size, num_mels=num_mels, prenet_dim=prenet_dim,
                            att_rnn_dim=att_rnn_dim, att_dim=att_dim, rnn_hidden_dim=rnn_hidden_dim,
                            postnet_dim=postnet_dim, dropout=dropout, scheduler_milestones=list(scheduler_milestones),
                            speaker_tokens=speaker_tokens, num_speakers=num_speakers, controls=controls,
                            controls_dim=controls_dim, max_len_override=max_len_override,
                            description_embeddings=description_embeddings,
                            description_embeddings_dim=description_embeddings_dim)
        self.lr, self.weight_decay = lr, weight_decay
        self.scheduler_milestones = list(scheduler_milestones)
        self.speaker_tokens, self.controls = speaker_tokens, controls
        self.max_len_override, self.description_embeddings = max_len_override, description_embeddings
        self.tacotron2 = Tacotron2(num_chars=num_chars, encoded_dim=encoded_dim, encoder_kernel_size=encoder_kernel_size,
                                   num_mels=num_mels, prenet_dim=prenet_dim, att_rnn_dim=att_rnn_dim, att_dim=att_dim,
                                   rnn_hidden_dim=rnn_hidden_dim, postnet_dim=postnet_dim, dropout=dropout,
                                   speaker_tokens=speaker_tokens, num_speakers=num_speakers, controls=controls,
                                   controls_dim=controls_dim, description_embeddings=description_embeddings,
                                   description_embeddings_dim=description_embeddings_dim, device=device)

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.tacotron2.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        cfg = {"optimizer": optimizer}
        if len(self.scheduler_milestones) > 0:
            sched = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=self.scheduler_milestones, gamma=0.1)
            cfg["lr_scheduler"] = {"scheduler": sched, "interval": "step"}
        return cfg

    def forward(self, chars_idx: Tensor, chars_idx_len: Tensor, teacher_forcing: bool = True,
                mel_spectrogram: Optional[Tensor] = None, mel_spectrogram_len: Optional[Tensor] = None,
                speaker_id: Optional[Tensor] = None, controls: Optional[Tensor] = None,
                max_len_override: Optional[int] = None, description_embeddings: Optional[Tensor] = None):
        return self.tacotron2(chars_idx=chars_idx, chars_idx_len=chars_idx_len, teacher_forcing=teacher_forcing,
                              mel_spectrogram=mel_spectrogram, mel_spectrogram_len=mel_spectrogram_len,
                              speaker_id=speaker_id, controls=controls, max_len_override=max_len_override,
                              description_embeddings=description_embeddings)

    def _args(self, meta):
        args = {}
        if self.speaker_tokens:
            args["speaker_id"] = meta["speaker_id"]
        if self.description_embeddings:
            args["description_embeddings"] = meta["description_embeddings"]
        if self.controls:
            args["controls"] = meta["features"]           # model/tts_model.py:125,173,304
        return args

    def _loss(self, batch):
        data, meta = batch[0], batch[1]
        mel, post, gate, alignment = self(chars_idx=data["chars_idx"], chars_idx_len=meta["chars_idx_len"],
                                          teacher_forcing=True, mel_spectrogram=data["mel_spectrogram"],
                                          mel_spectrogram_len=meta["mel_spectrogram_len"], **self._args(meta))
        # the three terms of model/tts_model.py:197-199 from the library's loss kernel (no ATen arithmetic on this path)
        l3 = _LossTermsFn.apply(mel, post, gate, data["mel_spectrogram"], data["gate"], meta["mel_spectrogram_len"])
        gate_loss, mel_loss, post_loss = l3[0], l3[1], l3[2]
        return l3.sum(), (gate_loss, mel_loss, post_loss), (mel, post, gate, alignment)

    def training_step(self, batch, batch_idx=0):
        return self._loss(batch)[0]

    def validation_step(self, batch, batch_idx=0):
        with torch.no_grad():
            loss, _, (mel, post, gate, alignment) = self._loss(batch)
        data, meta = batch[0], batch[1]
        ml, cl = meta["mel_spectrogram_len"], meta["chars_idx_len"]
        return {"mel_spectrogram_pred": post[0, :ml[0]], "mel_spectrogram": data["mel_spectrogram"][0, :ml[0]],
                "alignment": alignment[0, :ml[0], :cl[0]], "gate": data["gate"][0], "gate_pred": gate[0], "loss": loss}

    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        data, meta = batch[0], batch[1]
        with torch.no_grad():
            return self(chars_idx=data["chars_idx"], chars_idx_len=meta["chars_idx_len"], teacher_forcing=False,
                        max_len_override=self.max_len_override or 5000, **self._args(meta))

    # Lightning-style checkpoint exchange: {"state_dict": {"tacotron2.<name>": tensor}, "hyper_parameters": {...}}
    def checkpoint(self, extra: Optional[dict] = None) -> dict:
        ck = {"state_dict": {k: v.cpu() for k, v in self.tacotron2.state_dict(prefix="tacotron2.").items()},
              "hyper_parameters": dict(self.hparams)}
        ck.update(extra or {})
        return ck

    def load_checkpoint_dict(self, ck: dict, strict: bool = True):
        sd = {k[len("tacotron2."):]: v for k, v in ck["state_dict"].items() if k.startswith("tacotron2.")}
        return self.tacotron2.load_state_dict(sd, strict=strict)

    @classmethod
    def load_from_checkpoint(cls, path: str, map_location=None, device=None, **overrides):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        hp = dict(ck.get("hyper_parameters", {}))
        hp.update({k: v for k, v in overrides.items() if k in hp or k in ("lr", "weight_decay", "num_chars")})
        hp = {k: v for k, v in hp.items() if k in cls.__init__.__code__.co_varnames}
        model = cls(device=device, **hp)
        model.load_checkpoint_dict(ck)
        return model
