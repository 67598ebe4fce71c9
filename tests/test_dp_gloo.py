"""Data-parallel path on CPU with gloo, world_size 2 (the reference is single-device; DP is what this build adds).
Checks (a) the shard padding negotiation, (b) that ONE all-reduce of the flat gradient buffer followed by the 1/world scale
reproduces the single-process gradient of the concatenated batch (losses are plain means over padded tensors, so shards
are padded to the global (L, T)), and (c) that identical clip+Adam on every rank keeps replicas bit-identical.
Compute uses the CPU oracle (BatchNorm in eval mode: batch statistics are per shard by design)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tacotron2_ref as R
from tests.helpers import SMALL


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch():
    g = torch.Generator().manual_seed(7)
    lens = [9, 6, 12, 5]; tl = [11, 8, 13, 7]
    B, L, T, M = 4, max(lens), max(tl), SMALL["num_mels"]
    ci = torch.zeros(B, L, dtype=torch.int64); mel = torch.zeros(B, T, M); gate = torch.zeros(B, T, 1)
    for b in range(B):
        ci[b, :lens[b]] = torch.randint(1, 40, (lens[b],), generator=g)
        mel[b, :tl[b]] = torch.randn(tl[b], M, generator=g) - 3
        gate[b, :tl[b] - 1] = 1
    return ci, torch.tensor(lens), mel, torch.tensor(tl, dtype=torch.int32), gate


def _grads(P, d, ci, cl, mel, ml, gate):
    Pc = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not R.is_buffer(k) else v) for k, v in P.items()}
    o = R.tacotron2_fwd(Pc, d, ci, cl, True, mel, ml, training=False)
    loss = R.tts_loss(o[0], o[1], o[2], mel, gate)[0]
    names = [k for k, v in Pc.items() if v.requires_grad]
    return float(loss), dict(zip(names, torch.autograd.grad(loss, [Pc[k] for k in names])))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    d = R.default_dims(**SMALL, dropout=0.0)
    P = R.init_params(d, seed=3)
    ps = ParamStore(d, "cpu"); ps.load_state_dict(P)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
    assert tr.world == 2 and tr.rank == rank
    ci, cl, mel, ml, gate = _batch()
    sl = slice(2 * rank, 2 * rank + 2)                       # utterance sharding
    own = max(int(cl[sl].max()), 1), int(ml[sl].max())
    shard = dict(chars_idx=ci[sl, :own[0]].contiguous(), chars_idx_len=cl[sl], mel_spectrogram=mel[sl, :own[1]].contiguous(),
                 mel_spectrogram_len=ml[sl], gate=gate[sl, :own[1]].contiguous())
    shard = tr.global_pad(shard)                             # all_reduce(MAX) of (L, T)
    assert shard["chars_idx"].shape[1] == ci.shape[1] and shard["mel_spectrogram"].shape[1] == mel.shape[1]
    loss, g = _grads(P, d, shard["chars_idx"], shard["chars_idx_len"], shard["mel_spectrogram"],
                     shard["mel_spectrogram_len"], shard["gate"])
    ps.grad.zero_()
    for k, v in g.items():
        ps.G[k].copy_(v)
    dist.all_reduce(ps.grad)                                 # ONE collective on the flat buffer
    # replicated clip + Adam with grad_scale = 1/world (oracle restatement of t2_adam_step)
    gs = 1.0 / world
    coef, tot = R.clip_coef([ps.grad * gs], 1.0)
    newp, _, _ = R.adam_l2_step(ps.flat, ps.grad * gs * coef, torch.zeros_like(ps.flat), torch.zeros_like(ps.flat), 1, 1e-3, 1e-6)
    lt = torch.tensor([loss]); dist.all_reduce(lt)
    if rank == 0:
        out["loss"] = float(lt) / world
        out["grad"] = (ps.grad * gs).clone()
        out["offsets"] = dict(ps.offsets); out["shapes"] = {k: tuple(v) for k, v in ps.shapes.items()}
    h = torch.tensor([float(newp.double().sum()), float(newp.double().abs().sum())], dtype=torch.float64)
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    assert torch.equal(hs[0], hs[1]), "replicas diverged"
    dist.destroy_process_group()


def test_dp2_flat_allreduce_equals_single_process():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    d = R.default_dims(**SMALL, dropout=0.0)
    P = R.init_params(d, seed=3)
    ci, cl, mel, ml, gate = _batch()
    loss, g = _grads(P, d, ci, cl, mel, ml, gate)
    assert abs(out["loss"] - loss) < 1e-6 * max(1.0, abs(loss))
    for k, gr in g.items():
        o = out["offsets"][k]
        got = out["grad"][o:o + gr.numel()].view(out["shapes"][k])
        scale = max(float(gr.abs().max()), 1e-4)
        assert float((got - gr).abs().max()) / scale < 2e-4, k


# ---------------------------------------------------------------------------------------------------------------------------
# Shape negotiation in the loader thread (Trainer.negotiate_collated as DevicePrefetcher's `negotiate` hook)
# ---------------------------------------------------------------------------------------------------------------------------
class _FakeLoader:
    """`n` collated host batches per epoch with per-rank, per-batch lengths (data, metadata, extra) - the loader's layout."""

    def __init__(self, rank, n):
        self.rank, self.n, self.epoch = rank, n, 0

    def __len__(self):
        return self.n

    def __iter__(self):
        e = self.epoch
        self.epoch += 1
        for i in range(self.n):
            g = torch.Generator().manual_seed(1000 * self.rank + 10 * e + i)
            L = int(torch.randint(5, 20, (1,), generator=g)); T = int(torch.randint(8, 40, (1,), generator=g))
            data = dict(chars_idx=torch.ones(2, L, dtype=torch.int64), mel_spectrogram=torch.full((2, T, 4), -1.0),
                        gate=torch.ones(2, T, 1))
            meta = dict(chars_idx_len=torch.tensor([L, L - 1]), mel_spectrogram_len=torch.tensor([T, T - 2], dtype=torch.int32))
            yield data, meta, dict(own=(L, T))


def _prefetch_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tacotron2_amd.datasets.tts_dataset import DevicePrefetcher
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    d = R.default_dims(**SMALL, dropout=0.0)
    ps = ParamStore(d, "cpu")
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
    assert tr.dp and tr._shape_group is not None
    steps = 9                                                   # rank 0: epochs of 3 batches, rank 1: of 4 - nine steps cross both
    pf = DevicePrefetcher(_FakeLoader(rank, 3 + rank), lambda b, dev: b, "cpu", depth=2, negotiate=tr.negotiate_collated,
                          limit=steps, cycle=True)
    seen = []
    for data, meta, extra in pf:
        L, T = data["chars_idx"].shape[1], data["mel_spectrogram"].shape[1]
        assert data["gate"].shape[1] == T
        own = torch.tensor(extra["own"])
        both = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(both, own)                              # (main thread, default group: the data-path side)
        want = torch.stack(both).max(0).values
        assert (L, T) == (int(want[0]), int(want[1])), ((L, T), want)
        # the padding is zeros behind the rank's own data
        assert float(data["mel_spectrogram"][:, int(own[1]):].abs().sum()) == 0.0 and int(data["chars_idx"][:, int(own[0]):].sum()) == 0
        seen.append((L, T))
    assert len(seen) == steps and pf.produced == steps
    assert list(pf) == []                                       # the limit is spent: no further collective is entered
    out[rank] = seen
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_loader_thread_negotiates_the_global_shape_one_step_ahead():
    """Real data-parallel training takes its batches through DevicePrefetcher(negotiate=Trainer.negotiate_collated, limit, cycle):
    the global (L, T) of step k+1 is agreed by the loader threads over the trainer's host-side gloo group while step k runs, and
    train_step(padded=True) never sees a collective result.  Two ranks with epochs of different length (3 and 4 batches), nine
    steps: the k-th batches of both ranks always carry the same, maximal shape, zero padding behind the rank's own data, both
    threads end after exactly `limit` negotiations."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_prefetch_worker, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    assert out[0] == out[1] and len(out[0]) == 9


def _no_host_group_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import warnings
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    real_new_group = dist.new_group

    def failing_new_group(*a, **k):
        raise RuntimeError("no host transport (test)")
    dist.new_group = failing_new_group
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            tr = Trainer(ParamStore(R.default_dims(**SMALL, dropout=0.0), "cpu"), lr=1e-3, weight_decay=1e-6)
    finally:
        dist.new_group = real_new_group
    assert tr.dp and tr._shape_group is None and not tr.loader_negotiation
    assert any("main thread" in str(x.message) for x in w)
    # the loader-thread hook refuses (it would issue a collective of the data-path group from a second thread) ...
    data = dict(chars_idx=torch.ones(2, 5 + rank, dtype=torch.int64), mel_spectrogram=torch.zeros(2, 9 - rank, 4), gate=torch.zeros(2, 9 - rank, 1))
    with pytest.raises(RuntimeError, match="host-side"):
        tr.negotiate_collated((data, {}, {}))
    # ... and the main-thread negotiation still agrees the global shape, over the default group
    padded = tr.global_pad(data)
    out[rank] = (padded["chars_idx"].shape[1], padded["mel_spectrogram"].shape[1], padded["gate"].shape[1])
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_without_a_host_side_group_the_shape_is_agreed_on_the_main_thread_only():
    """when `dist.new_group(backend="gloo")` fails the loader thread must not negotiate (its MAX reduce would
    run on the gradient all-reduce's communicator from a second thread).  Trainer then reports loader_negotiation = False,
    negotiate_collated raises, run/train.py passes negotiate=None and train_step(padded=False) agrees the shape itself."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_no_host_group_worker, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    assert out[0] == out[1] == (6, 9, 9)


class _FakeWavBatch:
    """Stands in for datasets.tts_dataset.HostWavBatch: the host half of a training batch, whose padded lengths are host integers."""

    def __init__(self, L, T):
        self.L, self.T, self.Lg, self.Tg = L, T, L, T

    def set_global_shape(self, Lg, Tg):
        assert Lg >= self.L and Tg >= self.T
        self.Lg, self.Tg = Lg, Tg


def _wavbatch_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    tr = Trainer(ParamStore(R.default_dims(**SMALL, dropout=0.0), "cpu"), lr=1e-3, weight_decay=1e-6)
    shapes = [(17 + 5 * rank, 300 - 40 * rank), (9, 120 + 7 * rank), (30 - rank, 512)]
    got = []
    for L, T in shapes:
        hb = tr.negotiate_collated(_FakeWavBatch(L, T))        # the loader thread's hook, DeviceBatchLoader's batch type
        got.append((hb.Lg, hb.Tg))
    out[rank] = got
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_device_batch_loader_batches_negotiate_their_shape_from_host_integers():
    """A batch of the device-resident loader knows its padded (L, T) on the host (frames = 1 + samples // hop) before anything is on
    the device: Trainer.negotiate_collated agrees the step's global shape over the host-side group and only RECORDS it
    (set_global_shape); the batched log-mel pass then writes straight into a tensor of that shape."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_wavbatch_worker, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    assert out[0] == out[1] == [(22, 300), (9, 127), (30, 512)]
