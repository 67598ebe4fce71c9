"""Training-step time of the non-headline configurations at full size (GPU box only): BASELINE.json configs[3]
(descriptions + 562 speaker tokens, E' = 640) and the controls extension (5 controls, 4 speakers)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tacotron2_amd.init import init_parameters
from tacotron2_amd.params import ParamStore
from tacotron2_amd.synthetic import ljspeech_batch
from tacotron2_amd.trainer import Trainer

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
VARIANTS = {
    "vanilla (headline)": dict(bench.VANILLA),
    "descriptions + 562 speakers (E'=640)": dict(bench.VANILLA, num_speakers=562, description_embeddings=True, description_embeddings_dim=768),
    "controls (5) + 4 speakers": dict(bench.VANILLA, controls=True, controls_dim=5),
}
for name, dims in VARIANTS.items():
    ps = ParamStore(dims, dev)
    init_parameters(ps, seed=0)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6, scheduler_milestones=(50000, 75000))
    batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=dims["num_speakers"]).items()}
    g = torch.Generator().manual_seed(7)
    if dims.get("description_embeddings"):
        batch["description_embeddings"] = torch.randn(32, 768, generator=g).to(dev)
    if dims.get("controls"):
        batch["controls"] = torch.randn(32, 5, generator=g).to(dev)
    for _ in range(2):
        loss3, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss3, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    frames = float(batch["mel_spectrogram_len"].sum())
    print(f"{name:40s}: {dt * 1e3:7.2f} ms/step, {frames / dt / 1e3:6.1f} k valid mel-frames/s, loss {[round(float(x), 4) for x in loss3.cpu()]}", flush=True)
    del tr, ps
    torch.cuda.empty_cache()
