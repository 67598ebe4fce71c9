"""do_say (run/say.py:25-179): text -> ids -> checkpoint -> forward(teacher_forcing=False, max_len_override=5000) ON THE GPU
(the reference runs this on CPU at batch 1; here any number of texts is decoded as one batch, up to 64 per group) ->
log-mel written as .npy, or - when the output name ends in .wav - Griffin-Lim audio (tacotron2_amd/vocoder.py, the branch
run/say.py:161-171 takes without a HiFi-GAN checkpoint), or HiFi-GAN audio with --hifi-gan-checkpoint (tacotron2_amd/hifigan.py,
run/say.py:66-86,153-159)."""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch

from ..datasets.text import TextEncoder
from ..model.tts_model import TTSModel
from .common import model_kwargs


def load_description(description: Optional[str], dim: int, n: int, dev) -> torch.Tensor:
    """--description of `say`.  The reference encodes the description TEXT with google-bert/bert-base-uncased (run/say.py:93-116:
    tokenizer + BertModel -> pooler_output, (1, 768)) - remote weights, unavailable offline.  What the model consumes is only that
    vector, and the dataset path already feeds it from precomputed files (datasets/tts_dataset.py:277-287: torch.load of a `.pt`
    per utterance, zeros when absent).  So: a PATH to a precomputed embedding - `.pt` (weights-only load) or `.npy`, (dim,) or
    (1, dim) - is used as the dataset uses it; no description = zeros; raw text still needs BERT and is refused with that message."""
    import os
    if description is None:
        return torch.zeros(n, dim, device=dev)
    if not (os.path.isfile(description) and description.endswith((".pt", ".npy"))):
        raise NotImplementedError("--description with raw text needs the remote google-bert/bert-base-uncased weights (unavailable "
                                  "offline); pass the path of a precomputed pooler_output embedding (.pt or .npy, shape "
                                  f"({dim},)) instead, as the dataset manifests do")
    if description.endswith(".npy"):
        v = torch.from_numpy(np.load(description, allow_pickle=False))
    else:
        v = torch.load(description, map_location="cpu", weights_only=True)
    v = torch.as_tensor(v).float().reshape(-1)
    assert v.numel() == dim, f"--description embedding has {v.numel()} values, the model expects {dim}"
    return v.to(dev).unsqueeze(0).repeat(n, 1).contiguous()


def do_say(dataset_config: dict, training_config: dict, model_config: dict, extensions_config: dict, device: int,
           checkpoint: str, text: Union[str, List[str]], output: str, hifi_gan_checkpoint: Optional[str] = None,
           random_seed: Optional[int] = None, speaker_id: Optional[int] = None, controls: Optional[str] = None,
           description: Optional[str] = None, max_len: int = 5000):
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    pre = dataset_config["preprocessing"]
    enc = TextEncoder(pre["allowed_chars"], pre.get("end_token"), expand_abbrev=False)   # run/say.py: no abbreviations
    texts = [text] if isinstance(text, str) else list(text)
    ids = [torch.tensor(enc.encode(t), dtype=torch.int64) for t in texts]
    chars = torch.nn.utils.rnn.pad_sequence(ids, batch_first=True).to(dev)
    lens = torch.tensor([len(i) for i in ids], dtype=torch.int64, device=dev)
    cfg = dict(dataset=dataset_config, training=training_config, model=model_config, extensions=extensions_config)
    model = TTSModel.load_from_checkpoint(checkpoint, device=dev, **model_kwargs(cfg))
    model.eval()
    if random_seed is not None:
        model.tacotron2._seed = int(random_seed)
    kw = {}
    if model.speaker_tokens:
        kw["speaker_id"] = torch.full((len(texts),), int(speaker_id or 0), dtype=torch.int32, device=dev)
    if model.description_embeddings:
        dim = model.hparams["description_embeddings_dim"]
        kw["description_embeddings"] = load_description(description, dim, len(texts), dev)
    if model.controls:      # run/say.py:113-118: comma-separated values; the CLI help promises zeros by default
        n_ctl = int(model.hparams["controls_dim"])
        vals = [float(x) for x in controls.split(",")] if controls else [0.0] * n_ctl
        assert len(vals) == n_ctl, f"--controls needs {n_ctl} comma-separated values"
        kw["controls"] = torch.tensor([vals] * len(texts), dtype=torch.float32, device=dev)
    with torch.no_grad():
        _, post, gates, _ = model(chars_idx=chars, chars_idx_len=lens, teacher_forcing=False, max_len_override=max_len, **kw)
    post = post.cpu().numpy()
    # run/say.py:155,161 keeps mel_spectrogram_post[:, :-1]: all emitted frames but the last.  The stop frame itself is already
    # masked (gate -1000 from `lengths` on), so an utterance that stopped keeps its `lengths` = n - 1 frames; one that ran into
    # max_len without stopping has n unmasked frames and loses the last one, as in the reference.
    valid = (gates.cpu().numpy()[:, :, 0] != -1000.0).sum(1)
    n_emitted = post.shape[1]
    mels = [post[b, :max(min(int(valid[b]), n_emitted - 1), 1)] for b in range(len(texts))]
    if hifi_gan_checkpoint is not None:
        # run/say.py:66-86,153-159: generator(mel_post[:, :-1].swapaxes(1, 2)) -> waveform, written as audio whatever the name
        from ..hifigan import Generator
        from ..vocoder import write_wav
        sr = int(pre.get("sample_rate", 22050))
        gen = Generator.from_checkpoint(hifi_gan_checkpoint, device=dev)
        for i, m in enumerate(mels):
            wav = gen(torch.from_numpy(np.ascontiguousarray(m.T)).to(dev))[0, 0].cpu()
            name = output if isinstance(text, str) else f"{output.rsplit('.', 1)[0]}_{i}.wav"
            write_wav(name, wav, sr)
        return mels
    if output.endswith(".wav"):
        from ..vocoder import GriffinLim, write_wav
        sr = int(pre.get("sample_rate", 22050))
        gl = GriffinLim(n_mels=post.shape[2], sample_rate=sr, device=dev)
        for i, m in enumerate(mels):
            name = output if isinstance(text, str) else f"{output[:-4]}_{i}.wav"
            write_wav(name, gl.mel_to_audio(torch.from_numpy(m), seed=int(random_seed or 0)), sr)
        return mels
    np.save(output, mels[0] if isinstance(text, str) else np.array(mels, dtype=object), allow_pickle=not isinstance(text, str))
    return mels
