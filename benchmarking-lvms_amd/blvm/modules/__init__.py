from .convenience import AddConstant, Permute, View  # noqa: F401
from .distributions import DiagonalGaussianDense, DiscretizedLogisticMixtureDense  # noqa: F401
