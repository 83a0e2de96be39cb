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


class Compose(Transform):
    """Apply transforms left to right (transforms.py:30-52)."""

    def __init__(self, *transforms):
        super().__init__()
        self.transforms = nn.ModuleList([t for t in transforms if t is not None])

    def forward(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class RandomSegment(Transform):
    """A random window of `length` samples out of an example [T, *] (transforms.py:101-110); one `torch.randint` draw per
    call, shorter examples are returned whole."""

    def __init__(self, length: int):
        super().__init__()
        self.length = length

    def forward(self, x):
        start = int(torch.randint(low=0, high=max(x.size(0) - self.length, 1), size=(1,)))
        return x[start : start + self.length]


class Normalize(Transform):
    """(x - mean) / std with fixed statistics, or the example's own over `dim` (transforms.py:169-179)."""

    def __init__(self, mean=None, std=None, dim: int = -1):
        super().__init__()
        self.mean, self.std, self.dim = mean, std, dim

    def forward(self, x):
        mean = self.mean if self.mean is not None else x.mean(self.dim)
        std = self.std if self.std is not None else x.std(self.dim)
        return (x - mean) / std


class Denormalize(Transform):
    def __init__(self, mean=None, std=None):
        super().__init__()
        self.mean, self.std = mean, std

    def forward(self, x):
        return x * self.std + self.mean
