"""Collation of variable-length examples into padded batches, with the reference's names (blvm/data/batchers.py).
Host-side data plumbing: nothing here runs on the GPU."""
from typing import List, Optional

import torch


class Batcher:
    def __call__(self, batch):
        return self.collate(batch)

    def collate(self, batch):
        raise NotImplementedError()

    def sort(self, batch, sort_modality_idx: Optional[int] = None):
        return batch


class ListBatcher(Batcher):
    """Keep the examples as a python list (batchers.py:38-60)."""

    def collate(self, batch: List):
        return batch, torch.LongTensor([len(b) for b in batch])


class DynamicTensorBatcher(Batcher):
    """Right-pad tensors along one dynamic dimension to the longest example and stack them: [B, *, T_max, *] and the true
    lengths as a LongTensor (batchers.py:113-143); `sort` orders a batch longest first (:145-151)."""

    def __init__(self, dim: int = -1, pad_value: float = 0) -> None:
        self.dim, self.pad_value = dim, pad_value

    def collate(self, batch: List[torch.Tensor]):
        lengths = [t.shape[self.dim] for t in batch]
        shape = list(batch[0].shape)
        shape[self.dim] = max(lengths)
        out = torch.full([len(batch)] + shape, self.pad_value, dtype=batch[0].dtype)
        d = self.dim if self.dim >= 0 else batch[0].ndim + self.dim
        for i, (t, n) in enumerate(zip(batch, lengths)):
            out[i].narrow(d, 0, n).copy_(t)
        return out, torch.LongTensor(lengths)

    def sort(self, batch, sort_modality_idx: Optional[int] = None):
        if sort_modality_idx is None:
            return sorted(batch, key=lambda ex: len(ex[0]), reverse=True)
        return sorted(batch, key=lambda ex: ex[0][sort_modality_idx].shape[self.dim], reverse=True)
