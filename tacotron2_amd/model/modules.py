"""AlwaysDropout (reference model/modules.py:4-12): dropout that stays active in eval().  In this build dropout is a
scale mask handed to the kernels; this class only carries `p` and documents the always-on semantics."""
from torch import nn


class AlwaysDropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.training = True

    def forward(self, X):
        raise RuntimeError("AlwaysDropout is applied inside the fused prenet GEMM epilogue (mask from t2_philox_mask)")
