from .wavenet import InputSizeError, WaveNet  # noqa: F401
from .wavenet_modules import CausalConv1d, Conv1dResidualGLU, PointwiseTransform, ResidualStack  # noqa: F401
