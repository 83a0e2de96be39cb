"""Parameter-free shape / offset layers.  Only the constructor signatures are contract (the reference builds
`View(-1, n_batch_dims=2)` in front of its frame-stack encoders, blvm/models/vrnn.py:487, and `AddConstant(eps)` behind the
softplus of its Gaussian heads, blvm/modules/distributions.py:117); none of them owns a parameter, so state_dicts do not see them.
The HIP path never calls them: stacking is index arithmetic in the kernels, the epsilon is an argument of the head tile."""
import torch


class _Stateless(torch.nn.Module):
    """A module that is a pure function of its input and of a few constructor constants (shown by `extra_repr`)."""

    _fields = ()

    def extra_repr(self) -> str:
        return ", ".join(f"{name}={getattr(self, name)!r}" for name in self._fields)


class View(_Stateless):
    """Keep the first `n_batch_dims` axes, reshape the rest to `shape`."""

    _fields = ("shape", "n_batch_dims")

    def __init__(self, *shape: int, n_batch_dims: int = 1):
        super().__init__()
        self.shape, self.n_batch_dims = tuple(shape), int(n_batch_dims)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.view(x.shape[: self.n_batch_dims] + self.shape)  # a view, never a copy: incompatible strides raise


class Permute(_Stateless):
    _fields = ("dims",)

    def __init__(self, *dims: int):
        super().__init__()
        self.dims = tuple(dims)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.permute(x, self.dims)


class AddConstant(_Stateless):
    _fields = ("constant",)

    def __init__(self, constant: float):
        super().__init__()
        self.constant = constant

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.add(x, self.constant)
