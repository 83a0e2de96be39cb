"""Length-bucketed batch samplers with the reference's names, arguments and random-number consumption
(blvm/data/samplers/length_samplers.py:48-300), plus rank sharding for the data-parallel path: every rank builds the SAME
batches (same seed) and takes every `world_size`-th example of each, so one global batch is one optimizer step and the
per-rank frame counts feed `FlatGradAllReduce`'s exact normalisation.  Every rank must take the SAME number of steps (each
training step is a blocking all-reduce): a TRAINING batch with fewer examples than ranks is dropped on every rank alike; an
EVALUATION batch is never dropped — ranks beyond its size get an empty shard (`[]`), which the loops skip (evaluation has no
per-step collective; the metric sums are merged across ranks once, at the end).  Host-side plumbing."""
import csv
import random
from typing import Iterator, List, Optional, Union

import numpy as np


def load_field(source: str, field: str = "length") -> List[int]:
    """Column `field` of a source file: a CSV with a header whose column names may carry a suffix after a dot
    (`length.flac.samples`, prepare_timit.py:35) — the first column starting with `field` is used."""
    with open(source, newline="") as f:
        reader = csv.DictReader(f)
        col = next(c for c in reader.fieldnames if c == field or c.startswith(field + "."))
        return [int(float(row[col])) for row in reader]


def parse_max_len(batch_len, lengths) -> int:
    """`batch_len` as a number, or 'max' / '4max' = a multiple of the longest example (length_samplers.py:21-45)."""
    longest = int(max(lengths))
    if isinstance(batch_len, (int, float)):
        return int(batch_len)
    if isinstance(batch_len, str) and "max" in batch_len:
        digits = "".join(ch for ch in batch_len if ch.isdigit())
        return int(digits) * longest if digits else longest
    raise ValueError("`batch_len` must be an integer, float, or 'max'")


def _greedy_batches(order, lengths, budget):
    """Walk `order`, closing a batch whenever the next example would push its total length over `budget`."""
    batches, cur, total = [], [], 0
    for idx in order:
        n = lengths[idx]
        if total + n > budget:
            batches.append(cur)
            cur, total = [], 0
        cur.append(idx)
        total += n
    return batches, cur, total


class _Sharded:
    rank, world_size = 0, 1
    drop_small = False  # training: global batches smaller than the world are dropped identically on every rank

    def _shard(self, batch):
        return list(batch)[self.rank :: self.world_size] if self.world_size > 1 else batch

    def _global_batches(self):
        if self.drop_small and self.world_size > 1:
            return [b for b in self.batches if len(b) >= self.world_size]
        return self.batches

    def __len__(self):
        return len(self._global_batches())


class LengthTrainSampler(_Sharded):
    """Pools of similar-length examples; every epoch each pool is shuffled, batches of total length <= batch_len are cut
    greedily across the concatenated pools and the batch order is shuffled (length_samplers.py:48-194)."""

    drop_small = True

    def __init__(self, source: Union[str, List[int]], field: Optional[str] = "length", max_pool_difference: Optional[float] = None,
                 min_pool_size: int = 512, batch_len=None, batch_size=None, num_batches: Optional[int] = None, shuffle: bool = True,
                 longest_first: bool = True, drop_last: bool = True, rank: int = 0, world_size: int = 1):  # fmt: skip
        assert bool(batch_len) != bool(batch_size), "batch_len and batch_size are mutually exclusive."
        if not batch_len:
            raise NotImplementedError("`batch_size` is not yet implemented.")  # as the reference (:120)
        self.source, self.field = source, field
        self.min_pool_size, self.num_batches = min_pool_size, num_batches
        self.shuffle, self.longest_first, self.drop_last = shuffle, longest_first, drop_last
        self.batch_size, self.rank, self.world_size = batch_size, rank, world_size
        self.lengths = np.asarray(source if isinstance(source, list) else load_field(source, field), dtype=int)
        self.max_pool_difference = (self.lengths.max() - self.lengths.min()) * 0.05 if max_pool_difference is None else max_pool_difference
        self.sorted_indices = np.argsort(self.lengths)
        self.batch_len = parse_max_len(batch_len, self.lengths)
        self.buffer = []
        self.pools = self.create_sample_pools(self.max_pool_difference, min_pool_size)
        self.sample_batches()
        if longest_first:
            longest = max(range(len(self.batches)), key=lambda i: max(self.lengths[j] for j in self.batches[i]))
            self.batches[0], self.batches[longest] = self.batches[longest], self.batches[0]

    def create_sample_pools(self, max_diff, min_size):
        """Consecutive runs of the length-sorted examples: at least `min_size` examples, extended while the lengths stay
        within `max_diff` of the run's shortest; a tail shorter than `min_size` joins the last pool."""
        sorted_lens = self.lengths[self.sorted_indices]
        n, start, pools = len(sorted_lens), 0, []
        while start < n:
            base = sorted_lens[start]
            # the reference counts every example with base <= length < base + max_diff, including equal-length examples that
            # already went into the previous pool (length_samplers.py:146-148) — kept, pools then overshoot by those ties
            within = int(np.searchsorted(sorted_lens, base + max_diff, side="left") - np.searchsorted(sorted_lens, base, side="left"))
            end = min(start + max(min_size, within), n)
            if n - end < min_size:
                end = n
            pools.append(self.sorted_indices[start:end].tolist())
            start = end
        return pools

    def sample_batches(self):
        while True:
            if self.num_batches is not None and len(self.buffer) >= self.num_batches:
                self.batches, self.buffer = self.buffer[: self.num_batches], self.buffer[self.num_batches :]
                return
            order = np.concatenate([random.sample(p, k=len(p)) for p in self.pools])
            batches, last, last_len = _greedy_batches(order, self.lengths, self.batch_len)
            if last and not (self.drop_last and last_len < self.batch_len):
                batches.append(last)
            if self.shuffle:
                random.shuffle(batches)
            if self.num_batches is None:
                self.batches = batches
                return
            self.buffer += batches

    def __iter__(self) -> Iterator[List[int]]:
        try:
            for batch in self._global_batches():
                yield self._shard(batch)
        finally:
            if self.shuffle:
                self.sample_batches()


class LengthEvalSampler(_Sharded):
    """Deterministic batches over the length-sorted examples, bounded by total length or by count; longest batch first
    (length_samplers.py:197-300)."""

    def __init__(self, source: Union[str, List[int]], field: Optional[str] = "length", batch_len=None, batch_size: Optional[int] = None,
                 shuffle: bool = False, longest_first: bool = True, rank: int = 0, world_size: int = 1):  # fmt: skip
        assert bool(batch_len) != bool(batch_size), "batch_len and batch_size are mutually exclusive."
        self.source, self.field, self.batch_size = source, field, batch_size
        self.shuffle, self.longest_first, self.rank, self.world_size = shuffle, longest_first, rank, world_size
        self.lengths = np.asarray(source if isinstance(source, list) else load_field(source, field), dtype=int)
        self.sorted_indices = np.argsort(self.lengths)
        self.batch_len = parse_max_len(batch_len, self.lengths) if batch_len else None
        self.sample_batches()

    def sample_batches(self):
        if self.batch_len:
            batches, last, _ = _greedy_batches(self.sorted_indices, self.lengths, self.batch_len)
            if last:
                batches.append(last)
        else:
            idx = list(self.sorted_indices)
            batches = [idx[i : i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.longest_first:
            self.longest_first = not self.shuffle  # only on the first epoch when shuffling
            batches.reverse()
        elif self.shuffle:
            random.shuffle(batches)
        self.batches = batches

    def __iter__(self) -> Iterator[List[int]]:
        try:
            for batch in self.batches:
                yield self._shard(batch)
        finally:
            if self.shuffle:
                self.sample_batches()
