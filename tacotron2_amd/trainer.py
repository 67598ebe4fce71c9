"""Training step driver: forward + 3-term loss + backward + (data-parallel gradient all-reduce) + global-norm clip +
Adam(L2) + MultiStepLR, i.e. what Lightning does around TTSModel.training_step in the reference
(run/train.py:210-243, model/tts_model.py:78-91,165-253), on the HIP engine.

Data parallelism (new relative to the reference, which is single-device): one process per GPU, utterances sharded
across ranks, the flat fp32 gradient buffer all-reduced per step over RCCL/xGMI (torch.distributed backend "nccl") - ONE call
after the backward by default, or two buckets with the larger one started behind the backward frame loop (overlap_allreduce) -
then identical clip + Adam on every rank.  Shards are padded to the global (L, T) maxima so that the mean of the per-rank loss
means equals the single-device loss on the concatenated batch (the loss is a plain mean over padded tensors,
model/tts_model.py:197-199); the two lengths are agreed on the HOST, over a gloo group of the trainer's own, by the loader thread
while the previous step runs (negotiate_shape / negotiate_collated): train_step(padded=True) reads no collective result.
BatchNorm statistics are per shard unless sync_bn (the reference has no multi-device behaviour to match).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist

from .engine import Engine, zero_later
from .params import ParamStore


class Trainer:
    def __init__(self, ps: ParamStore, lr: float, weight_decay: float, scheduler_milestones: Sequence[int] = (),
                 max_norm: float = 1.0, seed: int = 1234, sync_bn: bool = False, overlap_allreduce: bool = False,
                 force_collectives: bool = False, shape_timeout_s: float = 300.0):
        self.ps = ps
        self.engine = Engine(ps)
        self.base_lr, self.weight_decay, self.max_norm = lr, weight_decay, max_norm
        self.milestones = sorted(int(m) for m in scheduler_milestones)
        self.global_step = 0
        self.seed = seed
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # dp: the step issues its collectives.  force_collectives runs them at world size 1 too (an initialised process group of ONE
        # rank): every RCCL call of the step - both gradient buckets, Work.wait(), the BatchNorm statistics - on the hardware a
        # builder with one GPU has (tests/test_gpu_rccl.py, bench.py --force-dp); the sums are identities there.
        self.dp = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        self.frozen: set = set()   # parameter names excluded from updates (fine-tuning, run/train.py:229-233)
        # sync_bn: BatchNorm batch statistics (8 layers, forward and backward) over ALL ranks' shards - 16 all-reduces of
        # 2C + 2 doubles per step - so that N x b utterances give the single-device result on the N*b batch (the reference's
        # BN layers see the whole batch, model/encoder.py:41, model/postnet.py:16,30,44).  Off: per-shard statistics.
        self.sync_bn = bool(sync_bn) and self.dp
        if self.sync_bn:
            self.engine.sync_bn_group = dist.group.WORLD
        # overlap_allreduce: the gradient buffer is reduced as two buckets.  The tail [prenet.0.weight, end) - 80 % of the bytes:
        # both decoder cells, attention, projection, postnet - is complete when the backward frame loop and its weight-gradient
        # GEMMs are; its all-reduce is started there (behind the engine's side stream) and runs next to the encoder backward
        # (BiLSTM recurrence + convolutions, ~2.6 ms of latency-bound launches).  The head (encoder, speaker table, description
        # linear: 22 MB) follows when the backward ends.  Same sums, same result as one all-reduce.
        # OFF by default (round 4): measured over RCCL at world size 1, where the collective is a device-local pass over the bucket,
        # the overlapped tail costs the encoder backward and the chains' tail 0.9 ms (63.6 against 62.7 ms per step for ONE call
        # after the backward, 62.5 without a process group) - it pays only if the exposed all-reduce of a real multi-GPU run is
        # longer than that, which nobody has measured.
        # Not with sync_bn: RCCL runs the collectives of one communicator in issue order on its own stream, so the eight small
        # BatchNorm reduces of the encoder backward would queue behind the 90 MB tail and stall the main stream - the overlap
        # would turn into a wait.  (The overlapped path has run over gloo at 2 and 4 ranks and over RCCL at world size 1, where the
        # stream ordering is what is exercised; no multi-GPU RCCL run exists - DESIGN.md section 7.)
        self.overlap_allreduce = bool(overlap_allreduce) and self.dp and not self.sync_bn
        self._tail_start = ps.offsets["prenet.0.weight"]
        self._tail_work = None
        if self.overlap_allreduce:
            self.engine.grad_tail_hook = self._start_tail_allreduce
        # Shape negotiation travels over its OWN host-side (gloo) group: the padded lengths of a batch are host integers, so no
        # device tensor is read and the data-path communicator sees nothing but gradient / statistics reduces - which also lets a
        # loader thread negotiate batch k+1 while step k runs (DevicePrefetcher(negotiate=...)); collectives of one communicator
        # must be issued in one order on every rank, two threads on the same group could not promise that.
        self._shape_group = None
        self.queue_check = None
        if self.dp:
            import datetime
            try:
                # (a short timeout: a rank that died leaves its peers' loader threads inside this group's all-reduce - they must
                #  get an error, not wait for the default half hour)
                self._shape_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=shape_timeout_s))
            except Exception as e:
                # No host transport.  The loader thread must then NOT negotiate (negotiate_collated is None below): the MAX reduce
                # would go over the data-path communicator from a second thread, next to the main thread's gradient reduces -
                # collectives of one communicator issued in different orders on different ranks.  The shape is agreed on the main
                # thread instead, in train_step(padded=False), on the default group (one host read per step).
                import warnings
                warnings.warn(f"Trainer: no gloo group for the shape negotiation ({e}); the padded shape of every step is agreed on "
                              "the main thread over the default process group (one host synchronisation per step)")
            if ps.device.type == "cuda":
                # The engine's two streams must run side by side (tacotron2_amd/__init__.py: GPU_MAX_HW_QUEUES); with a live
                # communicator in the process this is checked, not assumed - a step that silently serialises takes 86 instead of
                # 62 ms (profiles/r04_rccl_hw_queues.txt) and nothing else would say why.
                self.queue_check = self.engine.ensure_concurrent_streams()
                if not self.queue_check["ok"]:
                    import warnings
                    warnings.warn("Trainer: the engine's main and side streams do NOT run concurrently in this process "
                                  f"({self.queue_check}): every new side stream landed on the main stream's hardware queue.  Set GPU_MAX_HW_QUEUES=16 (or more) in the "
                                  "environment BEFORE the first GPU call of the process (import tacotron2_amd first); every "
                                  "training step will otherwise be ~35 % slower.")

    def _start_tail_allreduce(self):
        # (called by Engine.backward_tf with its side stream current: the collective is ordered behind that stream's work)
        self._tail_work = dist.all_reduce(self.ps.grad[self._tail_start:], async_op=True)

    def trainable_ranges(self):
        """[start, end) element ranges of the flat buffer that the optimizer updates: everything except the frozen tensors
        (adjacent ranges merged; alignment padding between tensors carries zeros and may be included)."""
        ps = self.ps
        if not self.frozen:
            return [(0, ps.numel)]
        names = list(ps.offsets)
        ranges = []
        for i, name in enumerate(names):
            if name in self.frozen:
                continue
            a = ps.offsets[name]
            b = ps.offsets[names[i + 1]] if i + 1 < len(names) else ps.numel
            if ranges and ranges[-1][1] == a:
                ranges[-1] = (ranges[-1][0], b)
            else:
                ranges.append((a, b))
        return ranges

    def lr_at(self, step: int) -> float:
        """MultiStepLR(gamma=0.1), stepped once per optimiser step (model/tts_model.py:83-88)."""
        return self.base_lr * (0.1 ** sum(1 for m in self.milestones if step >= m))

    def negotiate_shape(self, L: int, T: int):
        """Global (max over ranks) padded text and frame lengths of this step's shards: ONE tiny MAX all-reduce of two host integers
        on the host-side group.  Every rank must call it once per step, in step order (from one thread)."""
        if not self.dp:
            return int(L), int(T)
        if self._shape_group is not None:
            lt = torch.tensor([int(L), int(T)], dtype=torch.int64)
            dist.all_reduce(lt, op=dist.ReduceOp.MAX, group=self._shape_group)
        else:
            lt = torch.tensor([int(L), int(T)], dtype=torch.int64, device=self.ps.device)
            dist.all_reduce(lt, op=dist.ReduceOp.MAX)
            lt = lt.cpu()
        return int(lt[0]), int(lt[1])

    @staticmethod
    def pad_to(batch: dict, Lg: int, Tg: int) -> dict:
        """Zero-pad `chars_idx` (B, L) and `mel_spectrogram` / `gate` (B, T, .) of a batch dict to (Lg, Tg); host or device
        tensors (the padding the collate function applies, datasets/tts_dataloader.py:25-33, extended to the global shape)."""
        out = dict(batch)
        L = batch["chars_idx"].shape[1]
        T = batch["mel_spectrogram"].shape[1]
        if L < Lg:
            out["chars_idx"] = torch.nn.functional.pad(batch["chars_idx"], (0, Lg - L))
        if T < Tg:
            out["mel_spectrogram"] = torch.nn.functional.pad(batch["mel_spectrogram"], (0, 0, 0, Tg - T))
            out["gate"] = torch.nn.functional.pad(batch["gate"], (0, 0, 0, Tg - T))
        return out

    def global_pad(self, batch: dict) -> dict:
        """Pad this rank's shard to the global (L, T) maxima (no-op on one rank).  Host integers over the host-side group: no
        device tensor is read."""
        if not self.dp:
            return batch
        Lg, Tg = self.negotiate_shape(batch["chars_idx"].shape[1], batch["mel_spectrogram"].shape[1])
        return self.pad_to(batch, Lg, Tg)

    @property
    def loader_negotiation(self) -> bool:
        """True when the padded shape of a step may be agreed from the loader thread (a host-side group of the trainer's own
        exists); False: only on the main thread (train_step(padded=False) / global_pad)."""
        return self.dp and self._shape_group is not None

    def negotiate_collated(self, collated):
        """`negotiate` hook of DevicePrefetcher: a collated HOST batch (data, metadata, extra) of the loader, padded to the step's
        global shape in the loader thread - one step ahead of the training loop, which then calls train_step(padded=True)."""
        if not self.dp:
            return collated
        if hasattr(collated, "set_global_shape"):        # a HostWavBatch of DeviceBatchLoader: its (L, T) are host integers already
            collated.set_global_shape(*self.negotiate_shape(collated.L, collated.T))
            return collated
        data, meta, extra = collated
        if self._shape_group is None:
            raise RuntimeError("Trainer.negotiate_collated needs the host-side (gloo) group: without it the shape is agreed on the "
                               "main thread (train_step(padded=False)); see Trainer.loader_negotiation")
        Lg, Tg = self.negotiate_shape(data["chars_idx"].shape[1], data["mel_spectrogram"].shape[1])
        return self.pad_to(data, Lg, Tg), meta, extra

    def train_step(self, batch: dict, masks: Optional[dict] = None, padded: bool = False):
        """One optimisation step on this rank's shard.  Returns the device tensor loss3 = (gate, mel, post) means.
        padded=True: the caller has already brought the shard to the global (L, T) (global_pad) - the per-step shape
        negotiation, a tiny all-reduce followed by a host read, is skipped."""
        eng, ps = self.engine, self.ps
        if not padded:
            batch = self.global_pad(batch)
        ci, mel = batch["chars_idx"], batch["mel_spectrogram"]
        B, L = ci.shape
        T = mel.shape[1]
        eng.marks = []; eng.spans = []
        if masks is None:
            masks = eng.make_masks(B, L, T, True, self.seed + 7919 * self.rank, self.global_step)
        eng.mark("masks")
        outs, ctx = eng.forward_tf(ci, batch["chars_idx_len"], mel, batch["mel_spectrogram_len"],
                                   speaker_id=batch.get("speaker_id"),
                                   description_embeddings=batch.get("description_embeddings"), training=True, masks=masks,
                                   controls=batch.get("controls"))
        zero_later(ps.grad)       # (one region of the backward's first t2_zero_regions launch)
        loss3 = eng.loss_and_grads(outs, ctx, mel, batch["gate"])
        if self.dp:
            if self._tail_work is not None:       # two buckets: the tail has been in flight since the frame loop ended
                dist.all_reduce(ps.grad[:self._tail_start])
                self._tail_work.wait()            # (the current stream waits for the collective; no host block on RCCL)
                self._tail_work = None
            else:
                dist.all_reduce(ps.grad)          # ONE flat fp32 buffer over RCCL/xGMI
            eng.mark("allreduce")
        # Frozen tensors (requires_grad=False in the reference: grad None) take no part in the step: their gradients are
        # excluded from the global-norm clip (Lightning's clip_grad_norm_ skips them) and neither the parameters nor their
        # Adam moments move (torch.optim.Adam skips parameters without a gradient; no L2 term either).
        for name in self.frozen:
            zero_later(ps.G[name])
        self.global_step += 1
        # MultiStepLR: the k-th optimiser step (k = global_step, 1-based) runs after k-1 scheduler steps
        eng.adam_step(self.global_step, self.lr_at(self.global_step - 1), self.weight_decay, self.max_norm,
                      grad_scale=1.0 / self.world, ranges=self.trainable_ranges() if self.frozen else None)
        eng.mark("optimizer")
        return loss3, outs
