"""Metric accumulators with the reference's names, constructor arguments and merge rules
(blvm/evaluation/metrics.py:15-50 Metric, :117-157 LatestMeanMetric, :209-264 RunningMeanMetric, :365-468 subclasses).

Difference by design: the reference performs one `.sum().tolist()` device->host sync per metric (8 per VRNN step).
Here a model packs all its per-step sums into ONE device vector (`DeferredScalars`) that is fetched with a single
transfer the first time any metric value is read, so constructing metrics inside `forward` does not stall the stream.
"""
import math
from copy import deepcopy
from typing import List, Optional, Set, Union

import torch


class DeferredScalars:
    """A small device vector whose elements are read on the host lazily with one transfer."""

    def __init__(self, tensor: torch.Tensor):
        self._tensor = tensor.detach()
        self._values = None

    def values(self):
        if self._values is None:
            self._values = self._tensor.tolist()
            self._tensor = None
        return self._values

    def __getitem__(self, i):
        return _Deferred(self, i)

    def __deepcopy__(self, memo):
        out = DeferredScalars.__new__(DeferredScalars)
        out._tensor, out._values = None, list(self.values())
        return out


class _Deferred:
    def __init__(self, src: DeferredScalars, index: int, scale: float = 1.0):
        self.src, self.index, self.scale = src, index, scale

    def __float__(self):
        return self.src.values()[self.index] * self.scale

    def __mul__(self, k):
        return _Deferred(self.src, self.index, self.scale * k)

    __rmul__ = __mul__

    def __truediv__(self, k):
        return _Deferred(self.src, self.index, self.scale / k)

    def __neg__(self):
        return _Deferred(self.src, self.index, -self.scale)


def _total(v, default=None):
    """Sum of a tensor / float / deferred scalar as a float-convertible; None -> default."""
    if v is None:
        return default
    if isinstance(v, torch.Tensor):
        return v.detach().sum().tolist()
    return v


class Metric:
    base_tags = set()
    _str_value_fmt = "<.3"

    def __init__(self, name: str, tags: Set[str] = None, get_best: str = None, log_to_console: bool = True,
                 log_to_framework: bool = True):  # fmt: skip
        self.name = name
        self.tags = self.base_tags if tags is None else (tags | self.base_tags)
        self.get_best = GET_BEST[get_best] if get_best is not None else GET_BEST["none"]
        self.log_to_console = log_to_console
        self.log_to_framework = log_to_framework

    @property
    def value(self):
        raise NotImplementedError()

    @property
    def str_value(self):
        return f"{self.value:{self._str_value_fmt}f}"

    def update(self, metric):
        raise NotImplementedError()

    def copy(self):
        return deepcopy(self)

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self.name}, value={self.str_value})"


def min_value(metrics: List[Metric]):
    return min(metrics, key=lambda m: m.value)


def max_value(metrics: List[Metric]):
    return max(metrics, key=lambda m: m.value)


def no_value(metrics: List[Metric]):
    return None


GET_BEST = dict(none=no_value, min=min_value, max=max_value)


class LatestMeanMetric(Metric):
    def __init__(self, values, name: str, tags: Set[str] = None, reduce_by=None, get_best: str = None,
                 log_to_console: bool = True, log_to_framework: bool = True):  # fmt: skip
        super().__init__(name, tags, get_best, log_to_console, log_to_framework)
        numel = values.numel() if isinstance(values, torch.Tensor) else 1
        self._num = _total(values)
        self._den = _total(reduce_by) or numel
        self._latest = None

    @property
    def latest(self):
        if self._latest is None:
            self._latest = float(self._num) / float(self._den)
        return self._latest

    @property
    def value(self):
        return self.latest

    def update(self, metric: Metric):
        self._latest = metric.latest


class EMAMetric(Metric):
    """Exponential moving average of a mean (blvm/evaluation/metrics.py:160-206): `weight_by` is the weight of the NEW value
    on update."""

    def __init__(self, values, name: str, tags: Set[str] = None, reduce_by=None, weight_by=None, get_best: str = None,
                 log_to_console: bool = True, log_to_framework: bool = True):  # fmt: skip
        super().__init__(name, tags, get_best, log_to_console, log_to_framework)
        numel = values.numel() if isinstance(values, torch.Tensor) else 1
        self._num = _total(values)
        self._den = _total(reduce_by) or numel
        self._w = _total(weight_by)
        self._ema = None

    @property
    def weight_by(self):
        return float(self._w) if self._w is not None and float(self._w) != 0 else float(self._den)

    @property
    def ema(self):
        if self._ema is None:
            self._ema = float(self._num) / float(self._den)
        return self._ema

    @property
    def value(self):
        return self.ema

    def update(self, metric: "EMAMetric"):
        avg_weight = (self.weight_by + metric.weight_by) / 2
        self._ema = avg_weight * metric.ema + (1 - avg_weight) * self.ema


class RunningMeanMetric(Metric):
    def __init__(self, values, name: str, tags: Set[str] = None, reduce_by=None, weight_by=None, get_best: str = None,
                 log_to_console: bool = True, log_to_framework: bool = True):  # fmt: skip
        """`values` [B] (tensor), a float, or a deferred device scalar holding the SUM of the values;
        value = sum(values) / sum(reduce_by); merge weight = sum(weight_by) (defaults: numel, reduce_by)."""
        super().__init__(name, tags, get_best, log_to_console, log_to_framework)
        numel = values.numel() if isinstance(values, torch.Tensor) else 1
        self._num = _total(values)
        self._den = _total(reduce_by) or numel
        self._w = _total(weight_by)
        self._mean = None
        self._weight = None

    def _resolve(self):
        if self._mean is None:
            den = float(self._den)
            self._weight = float(self._w) if self._w is not None and float(self._w) != 0 else den
            self._mean = float(self._num) / den

    @property
    def running_mean(self):
        self._resolve()
        return self._mean

    @property
    def weight_by(self):
        self._resolve()
        return self._weight

    @property
    def value(self):
        return self.running_mean

    def update(self, metric: "RunningMeanMetric"):
        self._resolve()
        d = self._weight + metric.weight_by
        self._mean = self._mean * (self._weight / d) + metric.running_mean * (metric.weight_by / d)
        self._weight = d


class LossMetric(RunningMeanMetric):
    base_tags = {"losses"}

    def __init__(self, values, name: str = "loss", tags=None, reduce_by=None, weight_by=None, get_best="min", **kw):
        super().__init__(values, name, tags, reduce_by, weight_by, get_best, **kw)


class LLMetric(RunningMeanMetric):
    base_tags = {"log_likelihoods"}

    def __init__(self, values, name: str = "ll", tags=None, reduce_by=None, weight_by=None, get_best="max", **kw):
        super().__init__(values, name, tags, reduce_by, weight_by, get_best, **kw)


class KLMetric(RunningMeanMetric):
    base_tags = {"kl_divergences"}

    def __init__(self, values, name: str = "kl", tags=None, reduce_by=None, weight_by=None, get_best=None, **kw):
        super().__init__(values, name, tags, reduce_by, weight_by, get_best, **kw)


class BitsPerDimMetric(RunningMeanMetric):
    base_tags = set()
    _str_value_fmt = "<5.3"

    def __init__(self, values, name: str = "bpd", tags=None, reduce_by=None, weight_by=None, get_best="min", **kw):
        values = -(values.detach() if isinstance(values, torch.Tensor) else values) / math.log(2)
        super().__init__(values, name, tags, reduce_by, weight_by, get_best, **kw)
