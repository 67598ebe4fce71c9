#!/usr/bin/env python3
"""Headline benchmark: LJSpeech-shaped teacher-forced TRAINING throughput (mel-frames/s, whole job) of the
hand-written gfx950 path, one process per GPU (RCCL gradient all-reduce for N > 1).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under an external launcher (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this process IS one rank.  Without one (no WORLD_SIZE) it is the LAUNCHER: before any GPU call it starts N fresh child
processes of this same command line, one per rank (launch_ranks below), relays rank 0's JSON line and exits with the worst child's
return code.  `--dry-run-launch` prints the child commands and their environment instead of starting them.

One step = dropout-mask generation + forward + 3-term loss + backward + [all-reduce] + global-norm clip + Adam on one
synthetic batch of 32 utterances per GPU (SURVEY.md section 8d shapes, vanilla-lj-hifi-stop dims: 4 speaker tokens,
fp32, random-init weights).  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tacotron2_amd  # noqa: E402,F401  (first: pins GPU_MAX_HW_QUEUES before the HIP runtime starts - tacotron2_amd/__init__.py)

VANILLA = dict(num_chars=39, encoded_dim=512, encoder_kernel_size=5, num_mels=80, prenet_dim=256, att_rnn_dim=1024,
               att_dim=128, rnn_hidden_dim=1024, postnet_dim=512, dropout=0.5, speaker_tokens=True, num_speakers=4,
               description_embeddings=False, description_embeddings_dim=0)

# A PREDICTION, not a measurement (no multi-GPU run of this code exists yet: SCALE_r01..r04 were skipped): what the driver's
# 1/2/4/8-GPU curve should show, so that a shortfall is attributable: weak scaling, identical
# work per rank; per-rank step = the 1-GPU step (62.5 ms; 62.7 ms with a live RCCL communicator, measured at world size 1) + the
# 112.5 MB gradient all-reduce, exposed behind the backward (ONE call; ring over 7 xGMI links x ~153 GB/s: 2 * 7/8 * 112.5 MB per
# link pair = 1.3 ms at the link rate, 2.5 ms at half of it).
PREDICTED_SCALING = dict(ms_per_step_n1=62.5, ms_per_step_with_process_group=62.7,
                        exposed_allreduce_ms=dict(one_call="1.0-2.5 (112.5 MB: 1.3 ms as a ring at the 7-link xGMI rate)", two_buckets=0.3),
                        speedup=dict(n2=1.96, n4=3.9, n8=7.7), floor_speedup_n8=7.4,
                        note="value(N) / value(1) with ONE all-reduce call after the backward (the default); below 7.4 at N=8 means a rank "
                             "waits for another, the all-reduce runs below 45 GB/s, or GPU_MAX_HW_QUEUES is back at 4 (then 86 ms per step "
                             "on every rank: DESIGN.md section 7)")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak


def decoder_step_bytes(B, L, Ef, P=256, A=1024, D=1024, Ad=128, M=80, F=32, Kl=31):
    """Algorithmic bytes of ONE fused decoder step: weights once + per-sample streams."""
    w = (4 * A * (P + Ef + A) + 8 * A + 4 * D * (A + Ef + D) + 8 * D + A * Ad + Ad + F * 2 * Kl + Ad * F
         + (D + Ef) * M + M + (D + Ef) + 1)
    r = 4 * (L * Ad + L * Ef + 2 * L + 2 * A + 2 * D + Ef + P) + L
    wr = 4 * (2 * A + 2 * D + Ef + M + 1 + 2 * L)
    return 4 * w + B * (r + wr)


def decoder_step_flops(B, L, Ef, P=256, A=1024, D=1024, Ad=128, M=80, F=32, Kl=31):
    """Algorithmic flops of ONE autoregressive decoder step for B utterances (38.4 MFLOP per sample at
    L = 160 + the prenet): both LSTM cells, query projection, location conv + dense, energies, context, mel/stop projection."""
    per = (2 * 4 * A * (P + Ef + A) + 2 * 4 * D * (A + Ef + D) + 2 * A * Ad + L * (2 * 2 * Kl * F + 2 * F * Ad + 5 * Ad)
           + 2 * L * Ef + 2 * (D + Ef) * (M + 1) + 2 * (M * P + P * P))
    return B * per


def cpu_baseline(dims, batch, t_cap=200, b_cap=32, timed_steps=3, n_inf=100):
    """Oracle (CPU restatement pinned to the reference, oracle/tacotron2_ref.py) timed on this host's cores on a bounded
    sample of the same workload, by SURVEY.md section 8d's protocol: same synthetic batch, fp32, all of the process's cores,
    1 warm-up + 3 timed teacher-forced training steps (fwd + loss + bwd + clip + Adam) at B = 32 with the frames capped at 200,
    and 100 autoregressive inference steps at B = 32."""
    from oracle import tacotron2_ref as R
    # threads actually used: the process's CPU share (the GPU box gives 16 cores per GPU), never the machine total
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    d = R.default_dims(**dims)
    P = R.init_params(d, seed=0)
    names = [k for k, v in P.items() if v.is_floating_point() and not R.is_buffer(k)]
    ci = batch["chars_idx"][:b_cap]; cl = batch["chars_idx_len"][:b_cap]
    L = int(cl.max()); ci = ci[:, :L]
    tl = torch.clamp(batch["mel_spectrogram_len"][:b_cap], max=t_cap)
    T = int(tl.max())
    mel = batch["mel_spectrogram"][:b_cap, :T]; gate = batch["gate"][:b_cap, :T]
    spk = batch["speaker_id"][:b_cap] if "speaker_id" in batch else None
    g = torch.Generator().manual_seed(0)
    sm = lambda shape, p: (torch.rand(shape, generator=g) >= p).float() / (1 - p)
    Pn, M = d["postnet_dim"], d["num_mels"]
    m_ = {k: torch.zeros_like(P[k]) for k in names}
    v_ = {k: torch.zeros_like(P[k]) for k in names}

    def train_step(step):
        masks = dict(enc_drop=[sm((b_cap, L, d["encoded_dim"]), 0.5) for _ in range(3)],
                     prenet_drop=[sm((b_cap, T + 1, d["prenet_dim"]), 0.5) for _ in range(2)],
                     att_drop=sm((T, b_cap, d["att_rnn_dim"]), 0.1), dec_drop=sm((T, b_cap, d["rnn_hidden_dim"]), 0.1),
                     post_drop=[sm((b_cap, T, c), 0.5) for c in (Pn, Pn, Pn, Pn, M)])
        Pc = {k: (v.detach().requires_grad_(True) if k in m_ else v) for k, v in P.items()}
        new_stats = {}
        o = R.tacotron2_fwd(Pc, d, ci, cl, True, mel, tl, speaker_id=spk, training=True, masks=masks, new_stats=new_stats)
        loss = R.tts_loss(o[0], o[1], o[2], mel, gate)[0]
        grads = torch.autograd.grad(loss, [Pc[k] for k in names])
        coef, _ = R.clip_coef(list(grads), 1.0)
        with torch.no_grad():
            for k, gr in zip(names, grads):
                P[k], m_[k], v_[k] = R.adam_l2_step(P[k].detach(), gr * coef, m_[k], v_[k], step, 1e-3, 1e-6)
            P.update(new_stats)

    train_step(1)                                    # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    for s_ in range(timed_steps):
        train_step(2 + s_)
    dt = time.perf_counter() - t0
    frames = int(tl.sum()) * timed_steps
    # secondary metric on the CPU too: autoregressive inference steps of the same oracle, eval mode, the same utterances
    # (random weights never emit a stop, so exactly n_inf steps run)
    with torch.no_grad():
        Pi = {k: v.detach() for k, v in P.items()}
        t1 = time.perf_counter()
        R.tacotron2_fwd(Pi, d, ci, cl, False, speaker_id=spk, max_len_override=n_inf, training=False, masks={})
        dti = time.perf_counter() - t1
    return dict(value=frames / dt, unit="mel-frames/s", cores=cores, kind="port",
                sample=f"oracle/tacotron2_ref.py, 1 warm-up + {timed_steps} timed train steps (fwd+loss+bwd+clip+Adam), first "
                       f"{b_cap} utterances of the bench batch, frames capped at {t_cap} (L={L}, T={T}, {int(tl.sum())} valid "
                       f"frames per step), {dt:.1f} s timed",
                decode_steps_per_s=n_inf / dti,
                decode_sample=f"{n_inf} autoregressive steps, batch {b_cap}, includes encoder and postnet of the call, {dti:.1f} s")


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as N fresh child processes of this command line
    (one Popen per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), wait for all of them and
    return the worst return code.  Nothing here has touched the GPU (no HIP call, no torch.cuda.is_available()), and nothing is
    exec'ed: the children initialise the runtime themselves.  Rank 0 inherits this process's stdout (its ONE JSON line is the
    bench line); the other ranks' stdout goes to stderr.  A rank that fails takes the others down after a grace period (they would
    wait in a collective for ever): terminate, then kill, by the exact PIDs started here."""
    import subprocess
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    child_argv = [a for a in argv if a != "--dry-run-launch"]
    plan = []
    for r in range(n):
        env = dict(RANK=str(r), LOCAL_RANK=str(0 if args.share_gpu else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "16"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, cores // n))))
        cmd = [sys.executable, os.path.abspath(__file__)] + child_argv
        if os.environ.get("T2_BENCH_CHILD_CMD"):        # test hook (tests/test_bench_launcher.py): stand-in ranks
            cmd = os.environ["T2_BENCH_CHILD_CMD"].split()
        plan.append(dict(cmd=cmd, env=env))
    if args.dry_run_launch:
        print(json.dumps(dict(launch=plan, note="dry run: nothing started, no GPU call made")), flush=True)
        return 0
    if not args.share_gpu:
        have = torch.cuda.device_count()       # (counts devices without initialising the runtime on this image)
        if have < n:
            print(f"[bench] --gpus {n} but this node shows {have} GPU(s); use --share-gpu --backend gloo for a rehearsal on one card",
                  file=sys.stderr, flush=True)
            return 2
    from tacotron2_amd.build import build
    build(verbose=False)                       # hipcc, no GPU: once here instead of by rank 0 while the others wait in a barrier
    procs = []
    for r, p in enumerate(plan):
        procs.append(subprocess.Popen(p["cmd"], env=dict(os.environ, **p["env"]), cwd=ROOT,
                                      stdout=None if r == 0 else sys.stderr))
    print(f"[bench] launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}", file=sys.stderr, flush=True)
    deadline, first_fail, grace = (time.time() + args.launch_timeout if args.launch_timeout > 0 else None), None, 20.0
    # a launcher that is told to stop (the driver's time limit, Ctrl-C) takes its ranks with it - nothing is left running on the GPUs
    import signal
    stop = {"sig": None}

    def on_signal(signum, frame):
        stop["sig"] = signum
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        now = time.time()
        if stop["sig"] is not None and deadline is None:
            print(f"[bench] launcher: signal {stop['sig']}: ending the ranks", file=sys.stderr, flush=True)
            deadline = now - 1
        if first_fail is None and any(rc not in (None, 0) for rc in rcs):
            first_fail = now
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            print(f"[bench] launcher: rank(s) failed {bad}; the others get {grace:.0f} s", file=sys.stderr, flush=True)
        if (first_fail is not None and now - first_fail > grace) or (deadline is not None and now > deadline):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            if first_fail is None:
                print("[bench] launcher: --launch-timeout reached", file=sys.stderr, flush=True)
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    worst = next((rc for rc in rcs if rc != 0), 0)
    if worst:
        print(f"[bench] launcher: return codes per rank {rcs}", file=sys.stderr, flush=True)
    return worst if worst > 0 else (1 if worst else 0)       # (a signal-killed child: negative code -> 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the secondary autoregressive decode-rate measurement")
    ap.add_argument("--no-high", action="store_true", help="skip the extra steps at float32_matmul_precision=high (profiling runs)")
    ap.add_argument("--fixed-shape", action="store_true", help="every utterance L=160, T=860 (roofline accounting variant)")
    ap.add_argument("--matmul-precision", default="highest", choices=["highest", "high", "medium"],
                    help="the reference's training.float32_matmul_precision for the GEMMs; the judged line is 'highest' (fp32-exact)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--sync-bn", action="store_true", help="BatchNorm statistics over all ranks (training.sync_batchnorm)")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="the gradient all-reduce as two buckets, the larger one started behind the backward frame loop (default: ONE call "
                         "after the backward; at world size 1 over RCCL the overlapped variant measured 0.9 ms slower)")
    ap.add_argument("--one-allreduce", action="store_true", help="(the default; kept for old command lines)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel step (process group, both gradient all-reduce buckets, Work.wait, [sync-BN reduces]) "
                         "also at --gpus 1: every RCCL call of the step on a single-GPU box; the sums are identities")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="with --gpus N and no launcher: print the N child commands / environments as one JSON line and exit (no GPU call)")
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="launcher: seconds after which all ranks are ended (0: none)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run_launch):
        sys.exit(launch_ranks(args, sys.argv[1:]))     # this process is the launcher: it never touches the GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world} in the environment (an external launcher with another rank count); "
              "drop WORLD_SIZE to let bench.py start its own ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dp:        # (the process group first, before any GPU call of this process)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)

    from tacotron2_amd.build import build
    if rank == 0:
        build(verbose=False)                  # (a no-op when bench.py's own launcher has built already)
    if world > 1:
        dist.barrier()
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.synthetic import ljspeech_batch
    from tacotron2_amd.trainer import Trainer

    from tacotron2_amd.engine import set_float32_matmul_precision
    set_float32_matmul_precision(args.matmul_precision)
    ps = ParamStore(VANILLA, dev)
    init_parameters(ps, seed=0)           # identical replicas on every rank
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, scheduler_milestones=(50000, 75000), sync_bn=args.sync_bn,
                 overlap_allreduce=args.overlap_allreduce and not args.one_allreduce, force_collectives=args.force_dp)
    if args.share_gpu and world > 1:
        # several ranks on ONE card (rehearsals): a persistent launch needs all of its workgroups co-resident, which two processes
        # launching theirs at the same moment cannot promise each other (the bounded waits then time out and poison the step) -
        # the recurrences run as per-step launches here.  One rank per GPU (the real thing) keeps the persistent launches.
        tr.engine.dec_chain = "steps"; tr.engine.enc_chain = "steps"
    cpu_batch = ljspeech_batch(args.batch, seed=1234 + rank, num_speakers=4,
                               fixed_shape=(160, 860) if args.fixed_shape else None)
    batch = {k: v.to(dev) for k, v in cpu_batch.items()}
    batch = tr.global_pad(batch)
    B, L = batch["chars_idx"].shape
    T = batch["mel_spectrogram"].shape[1]
    frames = torch.tensor([float(batch["mel_spectrogram_len"].sum()), float(B * T)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(frames)

    def sync():
        if tr.dp:
            dist.barrier()
        torch.cuda.synchronize()

    # do the engine's two streams run side by side in THIS process (GPU_MAX_HW_QUEUES, tacotron2_amd/__init__.py)?  Trainer checks
    # it when data-parallel (a live communicator brings streams of its own); measured here at N = 1 too, reported in the line
    qc = tr.queue_check or tr.engine.ensure_concurrent_streams()

    # (the batch is fixed and already at the global shape: no per-step shape negotiation - it would put a host read of an
    #  all-reduce result in front of every step)
    for _ in range(args.warmup):
        tr.train_step(batch, padded=True)
    sync()
    tr.engine.profile = True
    step_marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss3, _ = tr.train_step(batch, padded=True)
        step_marks.append(tr.engine.marks)   # (train_step starts a new list per step; the events are read after the timed region)
    t_enq = time.perf_counter() - t0         # host: all steps enqueued (far below dt when the GPU is the bound, as it should be)
    sync()
    dt = time.perf_counter() - t0
    tr.engine.profile = False
    tr.engine.check_persistent_kernels()     # (after the timed region: a host read of the persistent launches' timeout flag)
    seg = tr.engine.segment_times_ms()       # last timed step (events recorded inside the timed region)
    # per rank, over all timed steps: GPU time of a step on the main stream (first to last mark), of which the gradient all-reduce
    # segment (RCCL: the wire AND the wait for the slowest rank's backward) - so that "a rank was slow" (its own_step_ms - allreduce_ms
    # is the largest) and "the wire was slow" (allreduce_ms large on every rank) can be told apart in a scaling run
    ar_ms, gpu_ms = [], []
    for marks in step_marks:
        if len(marks) >= 2:
            gpu_ms.append(marks[0][1].elapsed_time(marks[-1][1]))
            ar_ms.append(sum(e0.elapsed_time(e1) for (_, e0), (n1, e1) in zip(marks[:-1], marks[1:]) if n1 == "allreduce"))
    mine = dict(rank=rank, ms_per_step=dt / args.steps * 1e3, host_enqueue_ms_per_step=t_enq / args.steps * 1e3,
                gpu_step_ms=sum(gpu_ms) / max(len(gpu_ms), 1), allreduce_ms=sum(ar_ms) / max(len(ar_ms), 1),
                valid_frames=float(batch["mel_spectrogram_len"].sum()), queue_check_ok=(qc or {}).get("ok"))
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    # ... and the gradient all-reduce ALONE, behind a barrier (nobody waits for a slower rank's backward): the wire time of the 112.5 MB
    # buffer, to set against the `allreduce` segment of the step (wire + waiting)
    allreduce_alone = None
    if tr.dp:
        sync()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_reduce(ps.grad)
        sync()
        ea.record()
        for _ in range(5):
            dist.all_reduce(ps.grad)
        eb.record()
        torch.cuda.synchronize()
        t_ar = torch.tensor([ea.elapsed_time(eb) / 5], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
        nbytes = ps.numel * 4
        allreduce_alone = dict(ms=float(t_ar), bytes=nbytes, algbw_GBs=nbytes / (float(t_ar) * 1e-3) / 1e9,
                               busbw_GBs=(2 * (world - 1) / world if world > 1 else 0.0) * nbytes / (float(t_ar) * 1e-3) / 1e9)

    # Data-parallel runs also time the OTHER gradient all-reduce mode (one call after the backward <-> two buckets, the larger one started
    # behind the frame loop) for a few steps, outside the judged region: the default was chosen at world size 1, where the collective
    # is a device-local pass; a real multi-GPU run is the only place the choice can be measured.  Every rank takes part.
    other_mode = None
    if tr.dp and not tr.sync_bn:
        was = tr.overlap_allreduce
        tr.overlap_allreduce = not was
        tr.engine.grad_tail_hook = tr._start_tail_allreduce if tr.overlap_allreduce else None
        for _ in range(2):
            tr.train_step(batch, padded=True)
        sync()
        to0 = time.perf_counter()
        n_other = min(args.steps, 8)
        for _ in range(n_other):
            tr.train_step(batch, padded=True)
        sync()
        dto = torch.tensor([time.perf_counter() - to0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dto, op=dist.ReduceOp.MAX)
        other_mode = dict(allreduce="2 buckets, tail overlapped with the encoder backward" if tr.overlap_allreduce else "1 call after the backward",
                          ms_per_step=float(dto) / n_other * 1e3, steps=n_other,
                          note="the non-default all-reduce mode, timed beside the judged line (not the judged value)")
        tr.overlap_allreduce = was
        tr.engine.grad_tail_hook = tr._start_tail_allreduce if was else None

    # the same step with the GEMMs at the precision the reference's shipped configs ask for (training.float32_matmul_precision =
    # "high", run/train.py:170: three of the six bf16 partial products; mel L1 against the fp32 oracle stays < 1e-4,
    # tests/test_gpu_fullsize.py).  Reported beside the judged fp32-exact line, never as `value`.
    high = None
    if rank == 0 and world == 1 and args.matmul_precision == "highest" and not args.no_high:
        set_float32_matmul_precision("high")
        for _ in range(2):
            tr.train_step(batch, padded=True)
        torch.cuda.synchronize()
        th0 = time.perf_counter()
        nh = min(args.steps, 5)
        for _ in range(nh):
            tr.train_step(batch, padded=True)
        torch.cuda.synchronize()
        dth = time.perf_counter() - th0
        set_float32_matmul_precision("highest")
        high = dict(ms_per_step=dth / nh * 1e3, mel_frames_per_s=float(frames[0]) * nh / dth, steps=nh,
                    note="GEMMs at float32_matmul_precision=high (bf16x3, the reference configs' setting); not the judged value")

    # secondary metric of BASELINE.json ("decode steps/sec"): batched autoregressive inference, 64 utterances (configs[4]),
    # fixed 860 frames with the stop checks live (random weights never emit a stop), rank 0 only, outside the timed region
    decode = None
    decode_b1 = None
    if rank == 0 and world == 1 and not args.no_decode:      # (reported by the N = 1 run only: the other ranks would idle)
        ib = ljspeech_batch(64, seed=4321, num_speakers=4)
        eng = tr.engine
        ci, cl, spk = ib["chars_idx"].to(dev), ib["chars_idx_len"].to(dev), ib["speaker_id"].to(dev)
        eng.infer(ci, cl, 32, speaker_id=spk, training=False, seed=1)           # warm-up
        torch.cuda.synchronize()
        n_dec = 860     # a fixed 860 steps per utterance with the stop checks live
        eng.profile = True; eng.marks = []; eng.spans = []
        eng.mark("inf.start")
        t1 = time.perf_counter()
        eng.infer(ci, cl, n_dec, speaker_id=spk, training=False, seed=2, check_every=64)
        torch.cuda.synchronize()
        ddt = time.perf_counter() - t1
        eng.profile = False
        loop_ms = eng.segment_times_ms().get("inf.frame_loop", 0.0)      # HIP events around the frame loop only
        Ld = int(ci.shape[1])
        us_step = loop_ms * 1e3 / n_dec

        def decode_roofline(Bd, L_, us):
            """Both fractions of one autoregressive decoder step, and what actually bounds it.  Measured (round 4,
            profiles/r04_ab_cell_mfma_ablation.txt): with 3/4 of the cells' MFMAs removed the frame gets 1 % shorter - the step is
            NOT matrix-pipe bound; it is six dependent launches (each ~2-3 us of launch boundary + 1.3-3 us to the first dependent
            operand) whose K loops run at the rate a compute unit takes operands in (~65 GB/s per CU)."""
            gbs = decoder_step_bytes(Bd, L_, 512) / (us * 1e-6) / 1e9 if us > 0 else 0.0
            tfs = decoder_step_flops(Bd, L_, 512) / (us * 1e-6) / 1e12 if us > 0 else 0.0
            return dict(bound="latency: 6 dependent launches per frame + per-CU operand ingest (not the matrix pipe: "
                              "profiles/r04_ab_cell_mfma_ablation.txt)",
                        kernel=f"autoregressive decoder step, {Bd} utterance(s) (6 launches / frame)",
                        mfma_achieved_TFLOPs=tfs, mfma_peak_TFLOPs=MFMA_F32_PEAK_TFLOPS, mfma_frac=tfs / MFMA_F32_PEAK_TFLOPS,
                        hbm_achieved_GBs=gbs, hbm_peak_GBs=HBM_PEAK_GBS, hbm_frac=gbs / HBM_PEAK_GBS,
                        launch_floor_us=6 * 1.7, launch_floor_frac=6 * 1.7 / us if us > 0 else None,
                        algorithmic_bytes_per_step=decoder_step_bytes(Bd, L_, 512),
                        algorithmic_flops_per_step=decoder_step_flops(Bd, L_, 512))
        decode = dict(decode_steps_per_s=n_dec / ddt, utterance_frames_per_s=64 * n_dec / ddt, batch=64, frames=n_dec,
                      L=Ld, launches_per_frame=6, note="decode_steps_per_s includes encoder, conditioning and postnet of the call",
                      frame_loop_us_per_step=us_step, frame_loop_steps_per_s=1e6 / us_step if us_step > 0 else None,
                      roofline=decode_roofline(64, Ld, us_step))
        # the reference's own `say` shape (run/say.py:139-149): ONE utterance, the same 860 frames.  At one row the frame is pure
        # weight streaming (72.4 MB) - and six launch latencies.
        c1, l1_, s1 = ci[:1].contiguous(), cl[:1].contiguous(), spk[:1].contiguous()
        L1 = int(l1_[0]); c1 = c1[:, :L1].contiguous()
        eng.infer(c1, l1_, 32, speaker_id=s1, training=False, seed=1)
        torch.cuda.synchronize()
        eng.profile = True; eng.marks = []; eng.spans = []
        eng.mark("inf.start")
        tb1 = time.perf_counter()
        eng.infer(c1, l1_, n_dec, speaker_id=s1, training=False, seed=2, check_every=64)
        torch.cuda.synchronize()
        db1 = time.perf_counter() - tb1
        eng.profile = False
        us_b1 = eng.segment_times_ms().get("inf.frame_loop", 0.0) * 1e3 / n_dec
        decode_b1 = dict(batch=1, frames=n_dec, L=L1, decode_steps_per_s=n_dec / db1, frame_loop_us_per_step=us_b1,
                         frame_loop_steps_per_s=1e6 / us_b1 if us_b1 > 0 else None,
                         realtime_factor=(n_dec * 256 / 22050) / db1, roofline=decode_roofline(1, L1, us_b1),
                         note="the reference's `say` shape: one utterance, whole call (encoder + 860 frames + postnet); realtime_factor = "
                              "seconds of audio at 22.05 kHz / hop 256 per second of wall time")
        # SURVEY section 8d also asks for the utterances sorted by length: the same lock-step loop, reported beside.  (Runs in which
        # utterances stop at different frames are the parity tests' job: tests/test_gpu_fullsize.py, 64 utterances against the oracle.)
        order = torch.argsort(cl, descending=True)
        cis, cls_, spks = ci[order].contiguous(), cl[order].contiguous(), spk[order].contiguous()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eng.infer(cis, cls_, n_dec, speaker_id=spks, training=False, seed=2, check_every=64)
        torch.cuda.synchronize()
        decode["sorted_by_length_steps_per_s"] = n_dec / (time.perf_counter() - t2)

    if rank == 0:
        Ef = 512
        ms_step = dt / args.steps * 1e3
        value = float(frames[0]) * args.steps / dt
        # roofline of the teacher-forced decoder frame loop (north_star's "fused decoder step"): T decoder steps are
        # executed by {pre_att GEMM, attention chain, pre_dec GEMM, decoder-LSTM chain, projection GEMM}
        dec_fwd_ms = sum(v for k, v in seg.items() if k.startswith("fwd.dec."))
        alg = decoder_step_bytes(B, L, Ef) * T
        achieved = alg / (dec_fwd_ms * 1e-3) / 1e9 if dec_fwd_ms > 0 else 0.0
        out = dict(metric="mel-frames/sec (node), LJSpeech-shaped teacher-forced train step, b=32/GPU, fp32",
                   value=value, unit="mel-frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_step, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32" if args.matmul_precision == "highest" else f"f32 (GEMMs at float32_matmul_precision={args.matmul_precision})",
                   data="synthetic",
                   config=dict(workload="LJSpeech single-speaker train (vanilla-lj-hifi-stop.json dims), batch 32 per GPU, fp32",
                               global_batch=B * world, L=L, T=T, valid_frames_per_step=float(frames[0]),
                               padded_frames_per_step=float(frames[1]), parallelism=f"dp{world}",
                               sync_batchnorm=bool(tr.sync_bn),
                               allreduce=("none" if not tr.dp else "2 buckets, tail overlapped with the encoder backward"
                                          if tr.overlap_allreduce else "1 call after the backward"),
                               queue_check=qc,
                               hbm_footprint=dict(engine_workspaces=tr.engine.workspace_report(top=4),
                                                  parameters_grads_adam_bytes=4 * 4 * ps.numel,
                                                  torch_allocated_bytes=int(torch.cuda.max_memory_allocated(dev))),
                               shape_negotiation="none inside the timed region: one fixed batch, padded to the global shape before "
                                                 "it (main.py train agrees the shape of step k+1 on the host, in the loader thread, "
                                                 "over its own gloo group while step k runs: nothing on the step path either)",
                               predicted_unmeasured=PREDICTED_SCALING),
                   padded_frames_per_s=float(frames[1]) * args.steps / dt,
                   loss=[float(x) for x in loss3.cpu()],
                   roofline=dict(bound="hbm", kernel="teacher-forced decoder frame loop, forward (pre_att GEMM + attention "
                                 "chain + pre_dec GEMM + decoder-LSTM chain + projection GEMM = T fused decoder steps)",
                                 achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                                 traffic=None, algorithmic_bytes_per_step=decoder_step_bytes(B, L, Ef),
                                 us_per_decoder_step=dec_fwd_ms * 1e3 / T if T else None),
                   segments_ms={k: round(v, 3) for k, v in seg.items()},
                   per_rank=[{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in per_rank],
                   allreduce_alone=allreduce_alone, other_allreduce_mode=other_mode, decode=decode, decode_b1=decode_b1, matmul_precision_high=high)
        # HBM-side bytes of the same kernels from rocprofv3 PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE, profiles/):
        # recorded offline because counters cannot be collected inside this process
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            with open(tj) as f:
                tinfo = json.load(f)
            if tinfo.get("L") == L and tinfo.get("B") == B:
                out["roofline"]["traffic"] = tinfo["bytes_per_decoder_step_fwd"] * T
        if world == 1 and not args.no_cpu_baseline:
            dims = {k: v for k, v in VANILLA.items()}
            print("[bench] GPU timing done; timing the CPU oracle baseline (bounded sample)...", file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline(dims, cpu_batch, t_cap=200, b_cap=min(32, args.batch))
        print(json.dumps(out), flush=True)
    if tr.dp:
        dist.barrier()                       # every rank leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
