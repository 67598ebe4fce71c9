"""HiFi-GAN generator throughput at the UNIVERSAL_V1 shapes (random weights; GPU box): ms per utterance, x real time, TF/s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd.hifigan import Generator, UNIVERSAL_V1

dev = torch.device("cuda:0")
h = UNIVERSAL_V1
g = torch.Generator().manual_seed(0)
sd = {}
def conv(name, co, ci, k):
    sd[name + ".weight"] = torch.randn(co, ci, k, generator=g) * (1.0 / (ci * k) ** 0.5); sd[name + ".bias"] = torch.zeros(co)
def up(name, ci, co, k):
    sd[name + ".weight"] = torch.randn(ci, co, k, generator=g) * (1.0 / (ci * k) ** 0.5); sd[name + ".bias"] = torch.zeros(co)
ch = h["upsample_initial_channel"]
conv("conv_pre", ch, 80, 7)
flops_per_frame = 80 * ch * 7 * 2
Lmul = 1
for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
    up(f"ups.{i}", ch, ch // 2, k)
    flops_per_frame += Lmul * ch * (ch // 2) * k * 2
    ch //= 2; Lmul *= u
    for j, kk in enumerate(h["resblock_kernel_sizes"]):
        for c in range(3):
            conv(f"resblocks.{i * 3 + j}.convs1.{c}", ch, ch, kk); conv(f"resblocks.{i * 3 + j}.convs2.{c}", ch, ch, kk)
            flops_per_frame += 2 * Lmul * ch * ch * kk * 2
conv("conv_post", 1, ch, 7)
flops_per_frame += Lmul * ch * 7 * 2
gen = Generator(h, dev).load_state_dict(sd)
for T in (200, 860):
    mel = torch.randn(80, T, generator=g).to(dev)
    gen(mel); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        y = gen(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    secs = T * 256 / 22050
    print(f"T={T} frames ({secs:.1f} s of audio): {dt * 1e3:.1f} ms per utterance = {secs / dt:.0f} x real time, "
          f"{flops_per_frame * T / dt / 1e12:.1f} TF/s ({flops_per_frame * T / 1e9:.0f} GFLOP), finite={bool(torch.isfinite(y).all())}", flush=True)
