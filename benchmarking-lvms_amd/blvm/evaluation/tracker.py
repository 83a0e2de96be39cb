"""Minimal Tracker: the metric bookkeeping of the reference's training loops (blvm/evaluation/tracker.py:179-204
`steps`, :223-240 `epochs`, :377-392 `update`) without the terminal/wandb front-end (out of scope)."""
import time
from collections import defaultdict
from typing import Dict, Iterable, List

from .metrics import Metric


class Tracker:
    def __init__(self, print_every: float = None):
        self.metrics: Dict[str, Dict[str, Metric]] = defaultdict(dict)  # source -> name -> metric (current epoch)
        self.accumulated: Dict[str, Dict[str, List[Metric]]] = defaultdict(lambda: defaultdict(list))
        self.source = "train"
        self.epoch = 0
        self.step = 0
        self.print_every = print_every
        self._t0 = None

    def epochs(self, n: int):
        for e in range(1, n + 1):
            self.epoch = e
            self.metrics = defaultdict(dict)
            yield e

    def steps(self, loader: Iterable, source: str = None, max_steps: float = float("inf")):
        self.source = source or getattr(getattr(loader, "dataset", None), "source", None) or self.source
        self._t0 = time.time()
        for i, batch in enumerate(loader):
            if i >= max_steps:
                break
            self.step += 1
            yield batch

    __call__ = steps

    def update(self, metrics: List[Metric], source: str = None):
        """Merge by name with each metric's own rule (running mean weighted by `weight_by`, latest, ...)."""
        src = self.metrics[source or self.source]
        for m in metrics:
            if m.name in src:
                src[m.name].update(m)
            else:
                src[m.name] = m.copy()

    def all_reduce(self, source: str = None, group=None):
        """Data-parallel runs: merge this epoch's metrics of `source` across ranks with each metric's own rule (the reference's
        weighted running mean, blvm/evaluation/metrics.py:253-264), so that every rank — rank 0 decides about checkpoints — holds
        the value of the WHOLE evaluation set, not of its shard.  One host-side collective; a rank whose shards were all empty
        contributes nothing and still receives the result."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        src = self.metrics[source or self.source]
        for m in src.values():
            _ = m.value  # resolve deferred device scalars before pickling
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, {k: m.copy() for k, m in src.items()}, group=group)
        merged = {}
        for part in gathered:  # rank order: identical on every rank
            for name, m in part.items():
                if name in merged:
                    merged[name].update(m)
                else:
                    merged[name] = m.copy()
        src.clear()
        src.update(merged)

    def values(self, source: str = None):
        return {k: m.value for k, m in self.metrics[source or self.source].items()}

    def log(self):
        for src, ms in self.metrics.items():
            for name, m in ms.items():
                self.accumulated[src][name].append(m.copy())

    def best(self, source: str, name: str):
        ms = self.accumulated[source][name]
        return ms[0].get_best(ms) if ms else None
