"""Parameter initialisation with PyTorch's default distributions for the reference's layers (nn.Linear / nn.Conv1d:
kaiming-uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)); nn.LSTM / nn.LSTMCell: U(+-1/sqrt(hidden)); BatchNorm gamma 1,
beta 0) and the two N(0, 0.5) embeddings (model/encoder.py:26, model/tacotron2.py:65)."""
import math

import torch

from .params import ParamStore


def init_parameters(ps: ParamStore, seed: int = 0) -> None:
    g = torch.Generator().manual_seed(seed)
    for name, p in ps.P.items():
        shp = tuple(p.shape)
        if name in ("encoder.embedding.weight", "speaker_embedding.weight"):
            # .weight.data.normal_(0, 0.5) overwrites nn.Embedding's zeroed padding row too (model/encoder.py:25-26): row 0 is
            # NOT zero in the reference and feeds the encoder convolutions at padded positions
            v = torch.randn(shp, generator=g) * 0.5
        elif ("convolutions" in name or "postnet.postnet" in name) and len(shp) == 1 and int(name.split(".")[2]) % 4 == 1:
            v = torch.ones(shp) if name.endswith("weight") else torch.zeros(shp)   # BatchNorm affine
        elif "lstm" in name or "att_rnn" in name:
            bound = 1.0 / math.sqrt(shp[0] // 4)
            v = (torch.rand(shp, generator=g) * 2 - 1) * bound
        else:
            if len(shp) >= 2:
                fan_in = 1
                for x in shp[1:]:
                    fan_in *= x
            else:   # bias of a Linear/Conv: bound from the matching weight's fan_in
                w = ps.P.get(name[:-4] + "weight")
                fan_in = 1
                for x in (w.shape[1:] if w is not None else (shp[0],)):
                    fan_in *= x
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(shp, generator=g) * 2 - 1) * bound
        p.copy_(v.to(p.device))
    for name, b in ps.Bf.items():
        b.fill_(1.0 if name.endswith("running_var") else 0.0)
