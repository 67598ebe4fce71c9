"""Soak run of `main.py train`'s loop on generated WAVs (GPU box): N optimiser steps through the device-resident loader at vanilla dims,
B = 32, watching the allocator's footprint, the pinned-host footprint and the step rate per 100 steps - a leak or a drift would show.
usage: python tools/soak_train.py [steps]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tacotron2_amd  # noqa
import torch
from tools.bench_frontend import make_manifest


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    import bench, psutil
    from tacotron2_amd.datasets.tts_dataset import DeviceBatchLoader, DevicePrefetcher, TTSDataset
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="t2_soak_")
    try:
        files, texts = make_manifest(os.path.join(tmp, "wavs"), 1536)
        ds = TTSDataset(filenames=files, texts=texts, base_dir=os.path.join(tmp, "wavs"), silence=512, trim=True, cache=False, device=dev)
        ld = DeviceBatchLoader(ds, batch_size=32, shuffle=True, drop_last=True, seed=0, decode_threads=4)
        ps = ParamStore(dict(bench.VANILLA, speaker_tokens=False, num_speakers=1), dev); init_parameters(ps, 0)
        tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, scheduler_milestones=(steps // 2,))
        pf = DevicePrefetcher(ld, lambda b, d: b.to_device(d), dev, depth=2, limit=steps, cycle=True)
        proc = psutil.Process()
        t0 = time.perf_counter(); frames = 0; k = 0
        losses = []
        for b in pf:
            loss3, _ = tr.train_step(b, padded=True)
            k += 1
            if k % 100 == 0 or k == steps:
                l = [float(x) for x in loss3.cpu()]
                tr.engine.check_persistent_kernels()
                dt = time.perf_counter() - t0
                print(f"step {k:5d}  loss {sum(l):8.4f}  {100 / dt * 1e3 if k % 100 == 0 else 0:7.1f} steps/ks  {dt / 100 * 1e3 if k % 100 == 0 else 0:6.2f} ms/step  "
                      f"device allocated {torch.cuda.memory_allocated(dev) / 2**30:6.2f} GiB reserved {torch.cuda.memory_reserved(dev) / 2**30:6.2f} GiB  "
                      f"host RSS {proc.memory_info().rss / 2**30:5.2f} GiB  lr {tr.lr_at(k - 1):.1e}", flush=True)
                losses.append(sum(l))
                t0 = time.perf_counter()
        assert all(x == x for x in losses) and losses[-1] < losses[0], losses
        print("SOAK_OK", json.dumps(dict(steps=k, first_loss=losses[0], last_loss=losses[-1])))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
