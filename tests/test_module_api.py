"""Drop-in surface: class names, constructor kwargs, state_dict keys/shapes and checkpoint layout of the reference
(SURVEY.md section 8b, Appendix A).  CPU-only part: construction + (de)serialisation; the forward needs the GPU."""
import numpy as np
import pytest
import torch

from tests.helpers import SMALL, load_golden, params_from


def _small(**kw):
    from tacotron2_amd.model import Tacotron2
    return Tacotron2(dropout=0.5, device="cpu", **SMALL, **kw)


def test_state_dict_keys_and_shapes_match_reference():
    z = load_golden("tf_train_desc")
    m = _small(speaker_tokens=True, num_speakers=7, description_embeddings=True, description_embeddings_dim=24)
    sd = m.state_dict()
    ref = {k[2:]: v for k, v in z.items() if k.startswith("p.")}
    assert sorted(sd) == sorted(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(np.shape(v)), k
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in ref.items()})
    for k, v in ref.items():
        assert np.array_equal(m.state_dict()[k].numpy(), np.array(v)), k
    # named_parameters follow the reference's module paths
    names = dict(m.named_parameters())
    assert "decoder.attention.location_conv.weight" in names and "postnet.postnet.16.weight" in names


def test_checkpoint_roundtrip_lightning_layout(tmp_path):
    from tacotron2_amd.model import TTSModel
    kw = dict(lr=1e-3, weight_decay=1e-6, num_chars=39, char_embedding_dim=32, num_mels=16, prenet_dim=16, att_rnn_dim=32,
              att_dim=16, rnn_hidden_dim=32, postnet_dim=32, scheduler_milestones=[5, 7], device="cpu")
    a = TTSModel(**kw)
    ck = a.checkpoint()
    assert all(k.startswith("tacotron2.") for k in ck["state_dict"])
    assert ck["hyper_parameters"]["encoded_dim"] == 32          # char_embedding_dim alias (stale configs)
    path = tmp_path / "final.ckpt"
    torch.save(ck, path)
    b = TTSModel.load_from_checkpoint(str(path), device="cpu")
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    opt = b.configure_optimizers()
    assert isinstance(opt["optimizer"], torch.optim.Adam) and opt["lr_scheduler"]["interval"] == "step"


def test_forward_argument_checks_match_reference():
    m = _small()
    ci = torch.zeros(2, 5, dtype=torch.int64); cl = torch.tensor([5, 3])
    with pytest.raises(AssertionError):
        m(ci, cl, True)                                   # teacher forcing without mels
    with pytest.raises(Exception):
        m(ci, cl, False)                                  # no mels and no max_len_override
    with pytest.raises(RuntimeError):
        m(ci, cl, False, max_len_override=3)              # CPU tensors: there is no fallback path
