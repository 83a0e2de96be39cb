"""The two transforms on the hot path (reference: blvm/data/transforms.py:90-98 StackTensor, :192-215 µ-law)."""
import math

import torch
import torch.nn as nn

from blvm.utils.operations import stack_tensor


class Transform(nn.Module):
    def forward(self, x):
        raise NotImplementedError()


class StackTensor(Transform):
    def __init__(self, n_frames: int, dim=-1):
        super().__init__()
        self.n_frames = n_frames
        self.dim = dim

    def forward(self, x):
        x, _ = stack_tensor(x, self.n_frames, dim=self.dim)
        return x


class MuLawEncode(Transform):
    def __init__(self, bits: int = 8):
        super().__init__()
        self.bits = bits
        self.mu = 2**bits - 1
        self._divisor = math.log(self.mu + 1)

    def forward(self, x: torch.Tensor):
        return x.sign() * torch.log(1 + self.mu * x.abs()) / self._divisor


class MuLawDecode(Transform):
    def __init__(self, bits: int = 8):
        super().__init__()
        self.bits = bits
        self.mu = 2**bits - 1
        self._divisor = math.log(self.mu + 1)

    def forward(self, x: torch.Tensor):
        return x.sign() * (torch.exp(x.abs() * self._divisor) - 1) / self.mu
