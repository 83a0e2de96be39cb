from .metrics import (  # noqa: F401
    BitsPerDimMetric,
    DeferredScalars,
    EMAMetric,
    KLMetric,
    LatestMeanMetric,
    LLMetric,
    LossMetric,
    Metric,
    RunningMeanMetric,
)
from .tracker import Tracker  # noqa: F401
