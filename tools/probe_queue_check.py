"""Which probe sees the condition of profiles/r04_rccl_hw_queues.txt (GPU_MAX_HW_QUEUES=4 + a live RCCL communicator: 86 instead of
62 ms per step)?  Each case is a child process: [process group], several probe designs on the engine's two streams, then 5 timed
training steps at the bench shape as ground truth.   usage: python tools/probe_queue_check.py   (GPU box)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(case):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29689")
    os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
    sys.path.insert(0, ROOT)
    import tacotron2_amd  # noqa
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    if case == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    torch.cuda.set_device(0)
    import bench
    from tacotron2_amd._lib import call
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.synthetic import ljspeech_batch
    from tacotron2_amd.trainer import Trainer
    ps = ParamStore(bench.VANILLA, dev); init_parameters(ps, 0)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, force_collectives=(case == "nccl"))
    eng = tr.engine
    main, side = torch.cuda.current_stream(), eng.side_stream()
    w = torch.zeros(16, dtype=torch.int32, device=dev)
    first = eng.stream_concurrency_check()
    res = dict(case=case, queues=os.environ.get("GPU_MAX_HW_QUEUES"), A_first=first, A_ensured=eng.ensure_concurrent_streams())
    side = eng.side_stream()

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main); fn(); main.wait_stream(side); e1.record(main); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 1e3, 1)

    def pipeline(spin):      # the engine's pattern: a burst of dependent launches on main, event, the side stream waits and works
        def f():
            for _ in range(5):
                call("t2_stream_probe_chain", w, 192, main.cuda_stream)
                ev = main.record_event()
                side.wait_event(ev)
                if spin:
                    call("t2_stream_probe_spin", w.data_ptr() + 16, 1000, side.cuda_stream)
        return f
    res["D_pipeline_no_side_us"] = timed(pipeline(False))
    res["D_pipeline_side_spin5x1000_us"] = timed(pipeline(True))

    def both_chains():
        for _ in range(10):
            call("t2_stream_probe_chain", w, 50, main.cuda_stream)
            call("t2_stream_probe_chain", w.data_ptr() + 32, 50, side.cuda_stream)
    res["C_one_chain_500_us"] = timed(lambda: call("t2_stream_probe_chain", w, 500, main.cuda_stream))
    res["C_two_chains_500_us"] = timed(both_chains)
    batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=4).items()}
    for _ in range(2):
        tr.train_step(batch, padded=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.train_step(batch, padded=True)
    torch.cuda.synchronize()
    res["train_ms_per_step"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
    res["A_after_training"] = eng.stream_concurrency_check()
    print("PROBE " + json.dumps(res), flush=True)
    if case == "nccl":
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for q in ("4", "16", "1"):
            for c in ("none", "nccl"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), c], env=dict(os.environ, GPU_MAX_HW_QUEUES=q),
                                   capture_output=True, text=True, timeout=400)
                out = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")]
                print(out[-1] if out else f"{c} q={q}: FAILED {r.stderr[-800:]}", flush=True)
