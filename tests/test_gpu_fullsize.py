"""Oracle parity at PRODUCTION dims (vanilla-lj-hifi: E=512, prenet 256, H=1024, Ad=128, postnet 512, 80 mels), the
configuration bench.py measures.  The CPU oracle (oracle/tacotron2_ref.py, pinned to the reference's golden vectors)
runs the same seeded inputs at batch / sequence sizes it finishes in seconds; the HIP path goes through the C ABI.

Tolerances: mel L1 < 1e-4 (north_star) on mels / mels_post, alignments max-abs < 2e-5, loss 2e-5 relative, every
parameter gradient < 3e-4 relative to the tensor's scale (tests/test_gpu_model.py::_grad_check), BatchNorm running
statistics max-abs < 1e-5.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tacotron2_ref as R  # noqa: E402
from tests.helpers import dekink_masks  # noqa: E402
from tests.test_gpu_model import (MEL_L1_TOL, _dev, _grad_check, build_engine, l1, masks_to_device, mx,  # noqa: E402
                                  random_case)


def _ragged_case(d, B, L, T, seed, dev):
    ci, lens, mel, tl, gate, masks = random_case(d, B, L, T, seed, dev)
    return ci, lens, mel, tl, gate, masks


from tests.oracle_jobs import case as job_case, oracle_train as _oracle_train  # noqa: E402
from tests.oracle_pool import oracle, release  # noqa: E402


def _hip_train_and_compare(d, P, case, dev, kw_cpu=None, kw_dev=None, grad_tol=3e-4, check_engine=None, job=None):
    """job: name of the tests/oracle_jobs.py job that runs the oracle side of THIS case (same builder, same dekinked masks) in a
    background CPU process; None: the oracle runs here."""
    ci, lens, mel, tl, gate, masks = case
    if job is not None:
        o = oracle(job)
        ref, loss, grads, new_stats = o["ref"], o["loss"], o["grads"], o["new_stats"]
        masks = dict(masks, enc_drop=o["enc_drop"], prenet_drop=o["prenet_drop"])     # the (dekinked) masks the oracle ran with
        release(job)
    else:
        masks, _ = dekink_masks(P, d, ci, mel, masks)      # ReLU-kink elements out of both sides (tests/helpers.py)
        ref, loss, grads, new_stats = _oracle_train(P, d, ci, lens, mel, tl, gate, masks, **(kw_cpu or {}))
    eng, ps = build_engine(d, P, dev)
    if check_engine is not None:
        check_engine(eng)
    outs, ctx = eng.forward_tf(ci.to(dev), lens.to(dev), mel.to(dev), tl.to(dev), training=True,
                               masks=masks_to_device(masks, dev), **(kw_dev or {}))
    ps.grad.zero_()
    loss3 = eng.loss_and_grads(outs, ctx, mel.to(dev), gate.to(dev))
    torch.cuda.synchronize()
    eng.check_persistent_kernels()
    assert l1(outs[0], ref[0]) < MEL_L1_TOL and l1(outs[1], ref[1]) < MEL_L1_TOL, (l1(outs[0], ref[0]), l1(outs[1], ref[1]))
    assert mx(outs[0], ref[0]) < 1e-3 and mx(outs[1], ref[1]) < 2e-3
    assert mx(outs[3], ref[3]) < 2e-5
    assert ((outs[2].cpu() == -1000.0) == (ref[2] == -1000.0)).all()
    assert abs(float(loss3.sum()) - loss) < 2e-5 * max(1.0, abs(loss))
    _grad_check(ps, grads, tol=grad_tol)
    sd = ps.state_dict()
    for k, v in new_stats.items():
        if not k.endswith("num_batches_tracked"):
            assert mx(sd[k], v) < 1e-5, k
    return eng, ps


@pytest.mark.parametrize("B,L,T", [(3, 32, 16), (17, 29, 13)])
def test_vanilla_dims_train_step_matches_oracle(B, L, T):
    """configs[1] dims: forward outputs, loss, EVERY parameter gradient and the BN running statistics, ragged text and
    frame lengths; B = 17 spans two 16-row MFMA tiles of the step kernels."""
    dev = _dev()
    d = R.default_dims(speaker_tokens=True, num_speakers=4)       # vanilla-lj-hifi-stop.json: 4 speaker tokens
    P = R.init_params(d, seed=40 + B)
    case = _ragged_case(d, B, L, T, 500 + B, dev)
    spk = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(B), dtype=torch.int32)
    _hip_train_and_compare(d, P, case, dev, kw_cpu=dict(speaker_id=spk), kw_dev=dict(speaker_id=spk.to(dev)))


def test_vanilla_dims_train_step_at_matmul_precision_high():
    """training.float32_matmul_precision = "high" (run/train.py:170; what every shipped reference config sets): the GEMMs issue
    three of the six bf16 partial products.  Same production-dims step against the fp32 oracle at that precision class: the
    outputs stay inside the north-star mel tolerance, loss to 1e-4 relative, gradients to 3e-3 of each tensor's scale."""
    from tacotron2_amd import engine
    dev = _dev()
    d = R.default_dims(speaker_tokens=True, num_speakers=4)
    B, L, T = 3, 32, 16
    P = R.init_params(d, seed=40 + B)
    ci, lens, mel, tl, gate, masks = _ragged_case(d, B, L, T, 500 + B, dev)
    spk = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(B), dtype=torch.int32)
    ref, loss, grads, _ = _oracle_train(P, d, ci, lens, mel, tl, gate, masks, speaker_id=spk)
    try:
        engine.set_float32_matmul_precision("high")
        eng, ps = build_engine(d, P, dev)
        outs, ctx = eng.forward_tf(ci.to(dev), lens.to(dev), mel.to(dev), tl.to(dev), training=True,
                                   masks=masks_to_device(masks, dev), speaker_id=spk.to(dev))
        ps.grad.zero_()
        loss3 = eng.loss_and_grads(outs, ctx, mel.to(dev), gate.to(dev))
        torch.cuda.synchronize()
    finally:
        engine.set_float32_matmul_precision("highest")
    e0, e1 = l1(outs[0], ref[0]), l1(outs[1], ref[1])
    print(f"matmul precision high: mel L1 {e0:.2e} / {e1:.2e}, loss rel {abs(float(loss3.sum()) - loss) / abs(loss):.2e}")
    assert e0 < MEL_L1_TOL and e1 < MEL_L1_TOL, (e0, e1)
    assert abs(float(loss3.sum()) - loss) < 1e-4 * max(1.0, abs(loss))
    _grad_check(ps, grads, tol=3e-3)


def test_descriptions_libritts_dims_train_step_matches_oracle():
    """configs[3] dims: description embeddings (768 -> 128, E' = 640) + 562 speaker tokens."""
    dev = _dev()
    d = R.default_dims(speaker_tokens=True, num_speakers=562, description_embeddings=True, description_embeddings_dim=768)
    P = R.init_params(d, seed=71)
    B = 5
    case = _ragged_case(d, B, 27, 12, 571, dev)
    g = torch.Generator().manual_seed(9)
    spk = torch.randint(0, 562, (B,), generator=g, dtype=torch.int32)
    spk[1] = spk[0]                                              # two utterances of one speaker: the embedding gradient accumulates
    desc = torch.randn(B, 768, generator=g)
    _hip_train_and_compare(d, P, case, dev, kw_cpu=dict(speaker_id=spk, description_embeddings=desc),
                           kw_dev=dict(speaker_id=spk.to(dev), description_embeddings=desc.to(dev)))


def _default_schedule(eng):
    """The bench's schedule, untouched: 64-frame forward chunks with the persistent decoder-LSTM launches, 64-frame backward
    chunks with ramps, weight gradients in groups of four chunks, deferred weight-gradient GEMMs."""
    from tacotron2_amd.engine import _chunk_sizes
    assert (eng.chunk, eng.chunk_bwd, eng.dec_chain, eng.wgrad_group) == (64, 64, "persistent", 4)
    assert eng.ramp_chunks and eng.defer_wgrads and eng.chunk_att_wgrads and eng.bptt_off_chain
    assert _chunk_sizes(160, eng.chunk) == [64, 32, 32, 16, 8, 8]           # a full 64-frame persistent launch + the ramp
    assert _chunk_sizes(160, eng.chunk_bwd) == [64, 32, 32, 16, 8, 8]       # six backward chunks: two weight-gradient groups


@pytest.mark.oracle("bench_len_vanilla")
def test_vanilla_dims_bench_lengths_train_step_matches_oracle():
    """configs[1] dims AT THE BENCHMARKED LENGTHS: L = 188 characters (the bench batch's text length), T = 160 frames with the
    default schedule (so: attn_bwd_dw / attn_bwd_ds at Ad = 128, Ef = 512 and L = 188; the persistent decoder-LSTM chain at
    S = 64 steps and H = 1024; split-K weight gradients accumulated over two pipeline groups at H = 1024; the ramped chunks).
    Outputs, loss, EVERY parameter gradient and the BN running statistics against the oracle."""
    dev = _dev()
    c = job_case("bench_len_vanilla")            # B, L, T = 2, 188, 160 (utterance 0 the full text, the last one the full frame count)
    assert c["case"][0].shape == (2, 188) and c["case"][2].shape[1] == 160
    _hip_train_and_compare(c["d"], c["P"], c["case"], dev, kw_cpu=c["kw"], kw_dev={k: v.to(dev) for k, v in c["kw"].items()},
                           check_engine=_default_schedule, job="bench_len_vanilla")


@pytest.mark.oracle("bench_len_desc")
def test_descriptions_libritts_dims_bench_lengths_train_step_matches_oracle():
    """The same at configs[3] dims: E' = 640 (description embeddings) + 562 speaker tokens, L = 188, T = 160, default schedule
    (config/descriptions-libritts.json:21,42-52 of the reference)."""
    dev = _dev()
    c = job_case("bench_len_desc")
    assert c["d"]["num_speakers"] == 562 and c["case"][0].shape == (2, 188) and c["case"][2].shape[1] == 160
    _hip_train_and_compare(c["d"], c["P"], c["case"], dev, kw_cpu=c["kw"], kw_dev={k: v.to(dev) for k, v in c["kw"].items()},
                           check_engine=_default_schedule, job="bench_len_desc")


def test_descriptions_libritts_full_size_training_step_properties():
    """BASELINE configs[3] at its real per-GPU size: E' = 640, 562 speaker tokens, batch 64 over 2+ GPUs = 32 utterances per
    GPU, LibriTTS-shaped lengths (<= 10 s at 24 kHz: T <= 938; tacotron2_amd/synthetic.py).  No oracle at this size:
    attention rows are distributions supported on the text, masked tails are exactly 0 / -1000, BN running statistics move,
    every gradient is finite, the loss decreases on a fixed batch, the persistent launches report no timeout."""
    from tacotron2_amd.init import init_parameters
    from tacotron2_amd.params import ParamStore
    from tacotron2_amd.synthetic import ljspeech_batch
    from tacotron2_amd.trainer import Trainer
    dev = _dev()
    dims = dict(num_chars=39, encoded_dim=512, encoder_kernel_size=5, num_mels=80, prenet_dim=256, att_rnn_dim=1024,
                att_dim=128, rnn_hidden_dim=1024, postnet_dim=512, dropout=0.5, speaker_tokens=True, num_speakers=562,
                description_embeddings=True, description_embeddings_dim=768)
    ps = ParamStore(dims, dev); init_parameters(ps, 0)
    tr = Trainer(ps, lr=1e-3, weight_decay=1e-6)
    batch = {k: v.to(dev) for k, v in ljspeech_batch(32, seed=1234, num_speakers=562, desc_dim=768, shape="libritts").items()}
    assert batch["mel_spectrogram"].shape[1] > 872 and batch["mel_spectrogram"].shape[1] <= 938
    losses = []
    for step in range(4):
        loss3, (mels, post, gates, al) = tr.train_step(batch)
        losses.append(float(loss3.sum()))
        if step == 0:
            assert bool(torch.isfinite(ps.grad).all())
    torch.cuda.synchronize()
    tr.engine.check_persistent_kernels()
    assert all(np.isfinite(losses)) and losses[-1] < 0.6 * losses[0], losses
    cl, ml = batch["chars_idx_len"], batch["mel_spectrogram_len"]
    assert float((al.sum(-1) - 1).abs().max()) < 1e-4
    for b in (0, 7, 31):
        assert float(al[b, :, int(cl[b]):].abs().max()) == 0.0
        if int(ml[b]) < mels.shape[1]:
            assert float(mels[b, int(ml[b]):].abs().max()) == 0.0 and float(post[b, int(ml[b]):].abs().max()) == 0.0
            assert bool((gates[b, int(ml[b]):] == -1000.0).all())
    assert float((ps.Bf["postnet.postnet.1.running_mean"]).abs().max()) > 0
    # the speaker table only receives gradient in the rows of this batch's speakers
    gspk = ps.G["speaker_embedding.weight"]
    used = torch.zeros(562, dtype=torch.bool, device=dev); used[batch["speaker_id"].long()] = True
    assert float(gspk[~used].abs().max()) == 0.0 and float(gspk[used].abs().max()) > 0.0


def test_vanilla_dims_chunked_pipeline_matches_oracle():
    """The same production-dims step with the frame loop cut into several pipeline chunks (chunk 5 forward / 4 backward),
    so the chunked co-scheduled forward and the two-stream backward are exercised at H = 1024 against the oracle."""
    dev = _dev()
    d = R.default_dims()
    P = R.init_params(d, seed=3)
    B, L, T = 4, 24, 14
    ci, lens, mel, tl, gate, masks = _ragged_case(d, B, L, T, 77, dev)
    ref, loss, grads, _ = _oracle_train(P, d, ci, lens, mel, tl, gate, masks)
    eng, ps = build_engine(d, P, dev)
    eng.chunk, eng.chunk_bwd = 5, 4
    outs, ctx = eng.forward_tf(ci.to(dev), lens.to(dev), mel.to(dev), tl.to(dev), training=True, masks=masks_to_device(masks, dev))
    ps.grad.zero_()
    loss3 = eng.loss_and_grads(outs, ctx, mel.to(dev), gate.to(dev))
    torch.cuda.synchronize()
    assert l1(outs[0], ref[0]) < MEL_L1_TOL and l1(outs[1], ref[1]) < MEL_L1_TOL and mx(outs[3], ref[3]) < 2e-5
    assert abs(float(loss3.sum()) - loss) < 2e-5 * max(1.0, abs(loss))
    _grad_check(ps, grads)


@pytest.mark.parametrize("L,N,seed", [(40, 10, 64), (167, 12, 65)])
def test_vanilla_dims_batch64_inference_matches_oracle(L, N, seed):
    """configs[4]: autoregressive decoding of 64 variable-length utterances at production dims, ~10 frames, prenet
    dropout masks replayed, stop-logit bias nudged so utterances stop at different frames; frame count, masked tails,
    lengths and outputs against the oracle.  L = 167 is the text length of the bench's decode batch (SURVEY.md section 8d,
    seed 4321): the attention kernels of the decode loop at the benchmarked length."""
    dev = _dev()
    d = R.default_dims(speaker_tokens=True, num_speakers=4)
    P = R.init_params(d, seed=seed)
    P["decoder.gate.weight"] = P["decoder.gate.weight"] * 8.0      # stop logits cross zero at different frames (min |logit| 6e-4 / 4e-2)
    g = torch.Generator().manual_seed(seed)
    B = 64
    lens = torch.randint(9, L + 1, (B,), generator=g); lens[0] = L
    ci = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        ci[b, :lens[b]] = torch.randint(1, 40, (int(lens[b]),), generator=g)
    spk = torch.randint(0, 4, (B,), generator=g, dtype=torch.int32)
    pm = (torch.rand(N + 1, 2, B, 256, generator=g) >= 0.5).float() * 2
    masks = dict(prenet_drop=[[pm[i, 0], pm[i, 1]] for i in range(N + 1)])
    trace = {}
    with torch.no_grad():
        ref = R.tacotron2_fwd(P, d, ci, lens, False, speaker_id=spk, max_len_override=N, training=False, masks=masks, trace=trace)
    eng, ps = build_engine(d, P, dev)
    mels, post, gates, al, lengths = eng.infer(ci.to(dev), lens.to(dev), N, speaker_id=spk.to(dev),
                                               prenet_masks=pm.to(dev).contiguous(), check_every=4)
    torch.cuda.synchronize()
    assert mels.shape == ref[0].shape, (mels.shape, ref[0].shape)
    # the stop rule must have produced ragged lengths for this test to mean anything
    assert len(set(trace["lengths"].tolist())) > 1, trace["lengths"]
    assert (lengths.cpu() == trace["lengths"]).all()
    assert l1(mels, ref[0]) < MEL_L1_TOL and l1(post, ref[1]) < MEL_L1_TOL
    assert mx(al, ref[3]) < 5e-5
    assert ((gates.cpu() == -1000.0) == (ref[2] == -1000.0)).all()
    # bit-reproducible: the same call twice gives identical bits (no order-dependent reductions on the stop path)
    mels2, post2, gates2, al2, lengths2 = eng.infer(ci.to(dev), lens.to(dev), N, speaker_id=spk.to(dev),
                                                    prenet_masks=pm.to(dev).contiguous(), check_every=4)
    torch.cuda.synchronize()
    assert torch.equal(mels, mels2) and torch.equal(gates, gates2) and torch.equal(lengths, lengths2)


def _bn_ref64(x, gamma, beta, act, drop, eps=1e-5):
    """float64 BatchNorm1d (training statistics over all rows) + activation + dropout mask, with autograd."""
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    pre = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    y = pre
    if act == 1:
        y = torch.relu(pre)
    elif act == 2:
        y = torch.tanh(pre)
    return y * drop, mean, var, pre.detach()


@pytest.mark.parametrize("act", [1, 2])
def test_batchnorm_28k_rows_offset_input_matches_float64(act):
    """t2_bn_fwd / t2_bn_bwd at the bench's row count (B*T = 32*872 = 27,904 rows, C = 512) on an input with a large
    mean (-5.5, the log-mel level) and small spread: the variance must not lose digits to E[x^2] - E[x]^2 cancellation."""
    from tacotron2_amd._lib import call, make
    dev = _dev()
    B, L, C = 32, 872, 512
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, L, C, generator=g) * 0.25 - 5.5)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    drop = (torch.rand(B, L, C, generator=g) >= 0.5).float() * 2
    dy = torch.randn(B, L, C, generator=g) * 0.01
    x64 = x.double().reshape(B * L, C).requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y64, mean64, var64, pre64 = _bn_ref64(x64, g64, b64, act, drop.double().reshape(B * L, C))
    if act == 1:      # ReLU'(0) is a jump: take the elements within fp32 rounding of the kink out of the backward comparison
        dy = dy * (pre64.abs() > 1e-4).reshape(B, L, C).float()
    (y64 * dy.double().reshape(B * L, C)).sum().backward()

    Lp = L + 4
    xp = torch.zeros(B, Lp, C, device=dev)          # raw conv output in shifted row layout: row b*Lp + l
    xp_rows = xp.view(B * Lp, C)
    for b in range(B):
        xp_rows[b * Lp: b * Lp + L] = x[b].to(dev)
    y = torch.empty(B, Lp, C, device=dev)
    mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
    sums = torch.zeros(2 * C + 2, dtype=torch.float64, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gam, bet, drp = gamma.to(dev), beta.to(dev), drop.to(dev).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    bn = make("T2Bn", B=B, L=L, C=C, x=xp, Lp_x=Lp, gamma=gam, beta=bet, running_mean=rm, running_var=rv, training=1,
              momentum=0.1, eps=1e-5, sums=sums, mean=mean, invstd=invstd, act=act, drop=drp, y=y, Lp_y=Lp, pad_y=2)
    call("t2_bn_fwd", bn, st)
    torch.cuda.synchronize()
    n = B * L
    assert mx(mean, mean64.detach()) < 2e-6
    rel_istd = ((invstd.double().cpu() - 1 / torch.sqrt(var64.detach() + 1e-5)).abs() * torch.sqrt(var64.detach() + 1e-5)).max()
    assert float(rel_istd) < 2e-6, float(rel_istd)
    assert mx(rm, 0.1 * mean64.detach()) < 1e-6
    assert mx(rv, 0.9 + 0.1 * var64.detach() * n / (n - 1)) < 1e-6
    got = y[:, 2:2 + L].reshape(n, C)
    assert mx(got, y64.detach()) < 5e-4 and l1(got, y64.detach()) < 2e-5
    assert float(y[:, :2].abs().max()) == 0.0 and float(y[:, 2 + L:].abs().max()) == 0.0
    # backward
    dyd = dy.to(dev).contiguous()
    dx = torch.empty(B, Lp, C, device=dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    bnb = make("T2Bn", B=B, L=L, C=C, x=xp, Lp_x=Lp, gamma=gam, beta=bet, training=1, momentum=0.1, eps=1e-5, sums=sums,
               mean=mean, invstd=invstd, act=act, drop=drp, dy=dyd, Lp_dy=L, pad_dy=0, dx=dx, Lp_dx=Lp, pad_dx=2,
               dgamma=dgam, dbeta=dbet)
    call("t2_bn_bwd", bnb, st)
    torch.cuda.synchronize()
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert rel(dgam, g64.grad) < 1e-4 and rel(dbet, b64.grad) < 1e-4, (rel(dgam, g64.grad), rel(dbet, b64.grad))
    assert rel(dx[:, 2:2 + L].reshape(n, C), x64.grad) < 3e-4
