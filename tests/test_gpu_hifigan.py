"""HiFi-GAN generator on the GEMM kernels (model/hifi_gan.py:154-216, run/say.py:66-86,153-159)
against (a) the reference's own Generator outputs (tests/golden/hifigan.npz, oracle/make_golden_hifigan.py) and (b) the CPU
oracle at the UNIVERSAL_V1 layer shapes (kernel sizes 3/7/11, dilations 1/3/5, strides 8/8/2/2) with narrower channels.
Tolerance: waveform max-abs error 2e-5 (outputs are tanh-bounded in [-1, 1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import load_golden  # noqa: E402


@pytest.mark.parametrize("name", ["v1", "v2"])
def test_generator_matches_reference_fixture(name):
    from tacotron2_amd.hifigan import Generator
    dev = torch.device("cuda:0")
    z = load_golden("hifigan")
    sd = {k[len(name) + 4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith(name + ".sd.")}
    cfg = {k[len(name) + 5:]: v.tolist() for k, v in z.items() if k.startswith(name + ".cfg.")}
    g = Generator(cfg, dev).load_state_dict(sd)                       # weight_g / weight_v as in the published checkpoints
    wav = g(torch.from_numpy(z[name + ".mel"]).to(dev))
    torch.cuda.synchronize()
    ref = torch.from_numpy(z[name + ".wav"])
    assert tuple(wav.shape) == tuple(ref.shape)
    assert float((wav.cpu() - ref).abs().max()) < 2e-5


def test_generator_universal_v1_shapes_match_oracle():
    from oracle import hifigan_ref as H
    from tacotron2_amd.hifigan import UNIVERSAL_V1, Generator
    dev = torch.device("cuda:0")
    cfg = dict(UNIVERSAL_V1, upsample_initial_channel=64)            # 64 -> 32 -> 16 -> 8 -> 4 channels, every kernel shape of V1
    g = torch.Generator().manual_seed(2)
    rnd = lambda *s: torch.randn(*s, generator=g)
    sd = {"conv_pre.weight": rnd(64, 80, 7) * 0.05, "conv_pre.bias": rnd(64) * 0.1}
    ch = 64
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        sd[f"ups.{i}.weight"] = rnd(ch, ch // 2, k) * (0.6 / (ch * 2) ** 0.5); sd[f"ups.{i}.bias"] = rnd(ch // 2) * 0.1
        ch //= 2
        for j, (kk, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            for c in range(len(dil)):
                for part in ("convs1", "convs2"):
                    sd[f"resblocks.{i * 3 + j}.{part}.{c}.weight"] = rnd(ch, ch, kk) * (0.5 / (ch * kk) ** 0.5)
                    sd[f"resblocks.{i * 3 + j}.{part}.{c}.bias"] = rnd(ch) * 0.05
    sd["conv_post.weight"] = rnd(1, ch, 7) * 0.3; sd["conv_post.bias"] = rnd(1) * 0.1
    mel = rnd(2, 80, 11) * 1.5 - 4.0
    ref = torch.stack([H.generator_fwd({k: v.double() for k, v in sd.items()}, cfg, m.double()) for m in mel])
    wav = Generator(cfg, dev).load_state_dict(sd)(mel.to(dev))
    torch.cuda.synchronize()
    assert tuple(wav.shape) == (2, 1, 11 * 256)
    assert float(ref.abs().max()) > 0.05                              # a live signal, not a saturated or dead one
    assert float((wav[:, 0].double().cpu() - ref).abs().max()) < 2e-5


def test_unsupported_upsampler_shape_fails_loudly():
    from tacotron2_amd.hifigan import Generator
    cfg = dict(resblock="2", upsample_rates=[3], upsample_kernel_sizes=[7], upsample_initial_channel=8,
               resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3]])
    sd = {"conv_pre.weight": torch.zeros(8, 80, 7), "conv_pre.bias": torch.zeros(8), "ups.0.weight": torch.zeros(8, 4, 7),
          "ups.0.bias": torch.zeros(4)}
    with pytest.raises(NotImplementedError):
        Generator(cfg, "cuda:0").load_state_dict(sd)


def test_vocoding_a_padded_row_differs_from_vocoding_the_cut_row_only_in_the_tail():
    """run/test.py:167-181 vocodes the whole padded batch row (masked frames are zeros) and cuts the waveform at
    mel_length * 256.  The generator's receptive field spans many frames, so this is NOT the same as vocoding `mel[:n]`: the
    last samples see the activations of the zero frames that follow instead of zero margins.  tacotron2_amd/run/test.py follows
    the reference; this pins the reason (oracle on both inputs, reference fixture weights) and that the product reproduces the
    padded-row result."""
    from oracle import hifigan_ref as H
    from tacotron2_amd.hifigan import Generator
    z = load_golden("hifigan")
    name = "v1"
    sd = {k[len(name) + 4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith(name + ".sd.")}
    cfg = {k[len(name) + 5:]: v.tolist() for k, v in z.items() if k.startswith(name + ".cfg.")}
    mel = torch.from_numpy(z[name + ".mel"])[0]                                    # (80, T)
    n = mel.shape[1] // 2
    padded = mel.clone(); padded[:, n:] = 0.0
    up = int(np.prod(cfg["upsample_rates"]))
    folded = {k: v.double() for k, v in H.fold_weight_norm(sd).items()}
    ref_pad = H.generator_fwd(folded, cfg, padded.double())[:n * up]
    ref_cut = H.generator_fwd(folded, cfg, mel[:, :n].double())
    assert float((ref_pad - ref_cut).abs().max()) > 1e-2         # the two procedures differ (here, 6 frames, even everywhere)
    got = Generator(cfg, "cuda:0").load_state_dict(sd)(padded.cuda())[0, 0, :n * up].double().cpu()
    assert float((got - ref_pad).abs().max()) < 2e-5
