"""Why is the training step slower in a process that has initialised a torch.distributed process group over RCCL?  (GPU box.)
Each case is a child process: a preamble, then the SAME single-GPU training step without any collective (Trainer on one rank),
10 timed steps.  usage: python tools/probe_pg_slowdown.py            (parent: runs every case)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["none", "gloo", "nccl_eager", "nccl_lazy", "nccl_eager_allreduce", "nccl_eager_destroy"]
# (the package is imported AFTER the preamble here, so its GPU_MAX_HW_QUEUES default does not mask the effect being probed)


def child(case):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
    os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    if case == "gloo":
        dist.init_process_group("gloo")
    elif case.startswith("nccl_eager"):
        dist.init_process_group("nccl", device_id=dev)
    elif case == "nccl_lazy":
        dist.init_process_group("nccl")
    torch.cuda.set_device(0)
    if case == "nccl_eager_allreduce":
        x = torch.ones(1 << 20, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
    if case == "nccl_eager_destroy":
        x = torch.ones(1 << 20, device=dev); dist.all_reduce(x); torch.cuda.synchronize(); dist.destroy_process_group()
    sys.path.insert(0, ROOT)
    import bench
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.synthetic import ljspeech_batch
    from tacotron2_amd.trainer import Trainer
    ps = ParamStore(bench.VANILLA, dev); init_parameters(ps, 0)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)          # (no force_collectives: the step itself is identical in every case)
    assert not tr.dp
    batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
    for _ in range(3):
        tr.train_step(batch)
    torch.cuda.synchronize()
    tr.engine.profile = True
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    seg = tr.engine.segment_times_ms()
    env = {k: os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT", "GPU_MAX_HW_QUEUES", "HSA_ENABLE_SDMA", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING")}
    print(f"{case:24s} {dt:7.2f} ms/step  fwd chain {seg.get('fwd.dec.attn_chain', 0):6.2f}  bwd chains {seg.get('bwd.dec.chains', 0):6.2f}  env {env}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for extra_env in ({}, {"GPU_MAX_HW_QUEUES": "4"}, {"GPU_MAX_HW_QUEUES": "16"}):
            for c in CASES:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), c], env=dict(os.environ, **extra_env), capture_output=True, text=True, timeout=300)
                out = [l for l in r.stdout.splitlines() if "ms/step" in l]
                print((out[-1] if out else f"{c}: FAILED {r.stderr[-600:]}") + (f"   [{extra_env}]" if extra_env else ""), flush=True)
