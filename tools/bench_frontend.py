"""Throughput of the training input pipeline: does real data keep the GPU as busy as device-resident batches do?

Generates a manifest of noise WAVs with LJSpeech-shaped durations and texts (nothing to download: tacotron2_amd/synthetic.py's length
model, 16-bit PCM at 22.05 kHz), then, at batch 32 and vanilla dims, times
  * the loader alone (DeviceBatchLoader + DevicePrefetcher, no training): utterances/s delivered to the device, and the host part of it;
  * K training steps fed by the loader - cache off (the shipped configs' setting), cache cold (first epoch, entries written by the side
    copy), cache warm (every utterance from the cache), and the item-at-a-time loader (training.loader = "items");
  * the SAME K batches replayed from device memory (no loader at all): the reference point.
usage: python tools/bench_frontend.py [--utterances 1536] [--steps 40] [--out profiles/r05_frontend_throughput.txt]"""
import argparse, os, shutil, sys, tempfile, time, wave
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tacotron2_amd  # noqa
import numpy as np
import torch


def make_manifest(root, n_utt, sr=22050, seed=1234):
    """WAVs whose frame counts follow synthetic.ljspeech_batch's model: text length ~ clip(N(101, 33.6), 13, 188), frames =
    clip(5.68 len + N(0, 54), 99, 872); the signal is noise with a silent head and tail (so trimming has work to do)."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    files, texts = [], []
    letters = list("abcdefghijklmnopqrstuvwxyz ,.")
    for i in range(n_utt):
        ell = int(np.clip(round(rng.normal(101, 33.6)), 13, 188))
        t = int(np.clip(round(5.68 * ell + rng.normal(0, 54)), 99, 872))
        n = (t - 1) * 256 + int(rng.integers(0, 256)) - 512            # (512 samples of silence are appended by the dataset)
        x = (rng.standard_normal(n) * 0.1).astype(np.float32)
        head, tail = int(rng.integers(0, 4000)), int(rng.integers(0, 4000))
        x = np.concatenate([np.zeros(head, np.float32), x, np.zeros(tail, np.float32)])
        with wave.open(os.path.join(root, f"u{i}.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())
        files.append(f"u{i}.wav")
        texts.append("".join(rng.choice(letters, ell - 1)))
    return files, texts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=1536)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--decode-threads", type=int, default=4)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from tacotron2_amd.datasets.tts_dataset import DeviceBatchLoader, DevicePrefetcher, TTSDataLoader, TTSDataset
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.run.train import _to_dev
    from tacotron2_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="t2_frontend_")
    lines = []

    def say(s):
        print(s, flush=True); lines.append(s)
    try:
        t0 = time.time()
        files, texts = make_manifest(os.path.join(tmp, "wavs"), args.utterances)
        say(f"# manifest: {args.utterances} noise WAVs (LJSpeech-shaped durations, 16-bit PCM 22.05 kHz) written in {time.time() - t0:.1f} s; "
            f"batch {args.batch}, {args.steps} timed steps per case, vanilla dims, fp32, {args.decode_threads} decode threads, "
            f"{len(os.sched_getaffinity(0))} host cores")
        dims = dict(bench.VANILLA, speaker_tokens=False, num_speakers=1)
        ps = ParamStore(dims, dev); init_parameters(ps, 0)
        tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)

        def dataset(cache):
            return TTSDataset(filenames=files, texts=texts, base_dir=os.path.join(tmp, "wavs"), silence=512, trim=True, cache=cache,
                              cache_dir=os.path.join(tmp, "cache") if cache else None, device=dev)

        def loader_for(ds, kind):
            if kind == "items":
                return TTSDataLoader(ds, batch_size=args.batch, shuffle=True, drop_last=True), _to_dev
            return (DeviceBatchLoader(ds, batch_size=args.batch, shuffle=True, drop_last=True, seed=0, decode_threads=args.decode_threads),
                    lambda b, d: b.to_device(d))

        # ---- the loader alone -----------------------------------------------------------------------------------------------
        ds = dataset(False)
        ld, to_dev = loader_for(ds, "batched")
        pf = DevicePrefetcher(ld, to_dev, dev, depth=2, limit=args.steps)
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        for b in pf:
            nb += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        say(f"loader alone (batched, cache off): {nb} batches in {dt:.2f} s = {nb * args.batch / dt:.0f} utterances/s delivered to the device "
            f"(host decode + trim + pack: {ld.decode_s:.2f} s of it = {nb * args.batch / ld.decode_s:.0f} utterances/s on the loader thread)")

        # ---- training fed by the loader, and the same batches replayed from device memory ------------------------------------------
        warm = 3

        def run(kind, cache, tag):
            ds = dataset(cache)
            ld, to_dev = loader_for(ds, kind)
            pf = DevicePrefetcher(ld, to_dev, dev, depth=2, limit=args.steps + warm)
            kept, frames = [], 0
            it = iter(pf)
            for _ in range(warm):
                tr.train_step(next(it), padded=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for b in it:
                tr.train_step(b, padded=True)
                kept.append(b)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ds.flush_cache()
            frames = int(sum(int(b["mel_spectrogram_len"].sum()) for b in kept))
            # replay: the same batches, already on the device, no loader thread
            for b in kept[:warm]:
                tr.train_step(b, padded=True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for b in kept:
                tr.train_step(b, padded=True)
            torch.cuda.synchronize(); dr = time.perf_counter() - t1
            say(f"{tag:34s} {len(kept)} steps: {dt / len(kept) * 1e3:7.2f} ms/step {frames / dt:9.0f} frames/s | same batches from device memory: "
                f"{dr / len(kept) * 1e3:7.2f} ms/step {frames / dr:9.0f} frames/s | ratio {dr / dt:.3f}")
            return dt / len(kept), dr / len(kept)

        say("# case                               fed by the loader                     | replay (no loader)                     | replay / loader time")
        run("batched", False, "batched loader, cache off")
        run("batched", True, "batched loader, cache cold")
        n_cached = len(os.listdir(os.path.join(tmp, "cache")))
        run("batched", True, f"batched loader, cache warm*")
        say(f"#   (* {n_cached} of {args.utterances} utterances were in the cache when the warm run started)")
        run("items", False, "item-at-a-time loader, cache off")
        tr.engine.check_persistent_kernels()
        if args.out:
            with open(args.out, "w") as f:
                f.write("\n".join(lines) + "\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
