"""Shape helpers with the reference's names and semantics (blvm/modules/convenience.py:4-41)."""
import torch.nn as nn


class Permute(nn.Module):
    def __init__(self, *dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return x.permute(*self.dims)

    def __repr__(self):
        return f"Permute({self.dims})"


class View(nn.Module):
    def __init__(self, *shape, n_batch_dims: int = 1):
        super().__init__()
        self.shape = shape
        self.n_batch_dims = n_batch_dims

    def forward(self, x):
        return x.view(*x.shape[0 : self.n_batch_dims], *self.shape)

    def extra_repr(self):
        return f"{self.shape}, n_batch_dims={self.n_batch_dims}"


class AddConstant(nn.Module):
    def __init__(self, constant):
        super().__init__()
        self.constant = constant

    def forward(self, tensor1):
        return tensor1 + self.constant

    def __repr__(self):
        return f"AddConstant({self.constant})"
