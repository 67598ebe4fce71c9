"""The data-parallel training step over RCCL on ONE GPU: a process group of one rank (backend "nccl" = RCCL), started before any
GPU call of the process, Trainer(force_collectives=True).  Run by tests/test_gpu_rccl.py as a child process.

What it exercises - every RCCL call the 8-GPU job makes, on the transport the builder otherwise never reaches (all other DP tests
run over gloo): the tail bucket's all_reduce(async_op=True) issued under the engine's side stream, the head bucket's all-reduce,
Work.wait() on the main stream, the 16 small BatchNorm-statistics reduces of `--sync-bn`, barrier, destroy_process_group.  At
world size 1 a sum over ranks is the identity, which makes the stream ordering checkable bit for bit:
  * every reduced buffer is bit-identical to what was handed to the collective (clone taken at the call, on the calling stream);
  * the tail bucket covers exactly [offsets["prenet.0.weight"], end) and NOTHING writes to it after the hook: its content after
    the step equals the clone taken inside the hook;
  * loss, every gradient and the clip norm agree with the same step without collectives (split-K weight gradients use fp32
    atomics, so this comparison is to 1e-4 of each tensor's scale, not bitwise);
  * the whole step runs under torch.cuda.set_sync_debug_mode("error"): no call of the step blocks the host.
"""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tacotron2_amd  # noqa: E402,F401  (as every product entry point: pins GPU_MAX_HW_QUEUES before the HIP runtime starts)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))     # before any other GPU call
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from oracle import tacotron2_ref as R
    from tacotron2_amd import trainer as trainer_mod
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    from tests.test_gpu_model import masks_to_device, random_case
    sync_bn = "--sync-bn" in sys.argv
    overlap = "--one-allreduce" not in sys.argv
    d = R.default_dims(num_chars=39, encoded_dim=128, prenet_dim=64, att_rnn_dim=256, att_dim=64, rnn_hidden_dim=256,
                       postnet_dim=128, num_mels=80, dropout=0.5, speaker_tokens=True, num_speakers=4)
    P = R.init_params(d, seed=12)
    B, L, T = 6, 27, 150                       # three forward chunks (64 + ramp), several backward chunks
    ci, lens, mel, tl, gate, masks = random_case(d, B, L, T, 41, dev)
    spk = torch.tensor([0, 3, 1, 1, 2, 0], dtype=torch.int32)
    batch = dict(chars_idx=ci.to(dev), chars_idx_len=lens.to(dev), mel_spectrogram=mel.to(dev), mel_spectrogram_len=tl.to(dev),
                 gate=gate.to(dev), speaker_id=spk.to(dev))
    mdev = masks_to_device(masks, dev)
    ok, msg = True, []

    # ---- reference: the same step with no collective at all -------------------------------------------------------------
    ps0 = ParamStore(d, dev); ps0.load_state_dict(P)
    tr0 = Trainer(ps0, lr=1e-3, weight_decay=1e-6)
    assert not tr0.dp and tr0.world == 1
    loss0, _ = tr0.train_step(batch, masks=mdev)
    torch.cuda.synchronize()
    g0 = ps0.grad.clone()

    # ---- the data-parallel step over RCCL ---------------------------------------------------------------------------------
    ps = ParamStore(d, dev); ps.load_state_dict(P)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, sync_bn=sync_bn, overlap_allreduce=overlap, force_collectives=True)
    assert tr.dp and tr.world == 1 and tr.sync_bn == sync_bn and tr.overlap_allreduce == (overlap and not sync_bn)
    assert tr._shape_group is not None
    records = []                               # (tensor, clone at the call, Work or None, stream id)
    orig = dist.all_reduce

    def spy(t, *a, **kw):
        pre = t.clone()                        # on the calling stream, in front of the collective
        w = orig(t, *a, **kw)
        rec = dict(t=t, pre=pre, work=w if kw.get("async_op") else None, stream=torch.cuda.current_stream().cuda_stream,
                   numel=t.numel(), same=None)
        if not kw.get("async_op") and t.is_cuda:
            rec["same"] = (t == pre).all()     # enqueued behind the collective on the calling stream (no host read)
        records.append(rec)
        return w
    dist.all_reduce = spy
    trainer_mod.dist.all_reduce = spy
    try:
        tr.train_step(batch, masks=mdev, padded=True)        # warm-up: allocations, RCCL's lazy channel set-up
        torch.cuda.synchronize()
        ps.load_state_dict(P); ps.init_adam(); ps.exp_avg.zero_(); ps.exp_avg_sq.zero_(); tr.global_step = 0
        for k in list(ps.num_batches_tracked):
            ps.num_batches_tracked[k] = ps.num_batches_tracked[k] * 0
        records.clear()
        torch.cuda.set_sync_debug_mode("error")
        try:
            loss1, _ = tr.train_step(batch, masks=mdev, padded=True)
        except RuntimeError as e:
            ok = False; msg.append(f"a call inside the step synchronised the host: {e}")
            torch.cuda.set_sync_debug_mode("default")
            loss1, _ = tr.train_step(batch, masks=mdev, padded=True)
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    finally:
        dist.all_reduce = orig
        trainer_mod.dist.all_reduce = orig
    tr.engine.check_persistent_kernels()
    main_stream = torch.cuda.current_stream().cuda_stream
    grads = [r for r in records if r["t"].is_cuda and r["t"].dtype == torch.float32 and r["numel"] > 100000]
    stats = [r for r in records if r["t"].is_cuda and r["t"].dtype == torch.float64]
    tail0 = ps.offsets["prenet.0.weight"]
    if tr.overlap_allreduce:
        if len(grads) != 2:
            ok = False; msg.append(f"expected 2 gradient buckets, saw {len(grads)}")
        else:
            tail, head = grads
            if not (tail["work"] is not None and tail["numel"] == ps.numel - tail0 and tail["t"].data_ptr() == ps.grad.data_ptr() + 4 * tail0):
                ok = False; msg.append("tail bucket is not the async all-reduce of [prenet.0.weight, end)")
            if not (head["work"] is None and head["numel"] == tail0 and head["t"].data_ptr() == ps.grad.data_ptr()):
                ok = False; msg.append("head bucket is not [0, prenet.0.weight)")
            if tail["stream"] == main_stream or head["stream"] != main_stream:
                ok = False; msg.append("the tail must be issued under the side stream, the head under the main stream")
            if not torch.equal(ps.grad[tail0:], tail["pre"]):
                ok = False; msg.append("the tail bucket changed after the hook (late write, or the reduce is not an identity)")
            if not bool(head["same"]):
                ok = False; msg.append("head bucket: reduce at world 1 is not an identity")
    else:
        if len(grads) != 1 or grads[0]["numel"] != ps.numel or not bool(grads[0]["same"]):
            ok = False; msg.append("expected ONE identity all-reduce of the whole flat gradient buffer")
    if sync_bn:
        if len(stats) != 16 or not all(bool(r["same"]) for r in stats):
            ok = False; msg.append(f"sync-BN: expected 16 identity reduces of the statistics, saw {len(stats)}")
    elif stats:
        ok = False; msg.append("BatchNorm statistics were reduced without sync_bn")
    # ---- against the step without collectives ---------------------------------------------------------------------------------
    dl = abs(float(loss1.sum()) - float(loss0.sum()))
    if not dl < 1e-6 * max(1.0, float(loss0.sum())):
        ok = False; msg.append(f"loss differs by {dl}")
    worst = ("", 0.0)
    for name in ps.P:
        o, n = ps.offsets[name], ps.P[name].numel()
        r = g0[o:o + n].double()
        err = float((ps.grad[o:o + n].double() - r).abs().max()) / max(float(r.abs().max()), 1e-3)
        if err > worst[1]:
            worst = (name, err)
    msg.append(f"worst gradient mismatch against the no-collective step {worst[1]:.2e} at {worst[0]}")
    if not worst[1] < 1e-4:
        ok = False
    n0, n1 = float((g0.double() ** 2).sum().sqrt()), float((ps.grad.double() ** 2).sum().sqrt())
    if not abs(n0 - n1) < 1e-5 * n0:
        ok = False; msg.append(f"gradient norm {n1} vs {n0}")
    for k in ps.Bf:
        e = float((ps.Bf[k] - ps0.Bf[k]).abs().max())
        if not e < 1e-6:
            ok = False; msg.append(f"running statistic {k} differs by {e}")
    # ---- shape negotiation + timing of the collectives ---------------------------------------------------------------------
    if tr.negotiate_shape(11, 7) != (11, 7):
        ok = False; msg.append("negotiate_shape at world 1 is not the identity")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        tr.train_step(batch, masks=mdev, padded=True)
    ev[1].record(); torch.cuda.synchronize()
    msg.append(f"{ev[0].elapsed_time(ev[1]) / 5:.2f} ms per step with collectives (mid dims)")
    print(("RCCL_CHECK_OK " if ok else "RCCL_CHECK_FAIL ") + "; ".join(msg), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
