"""Source-file datasets with the reference's interface (blvm/data/base_dataset.py:17-166): a CSV whose first column is the
example's file stem (`filename,length.<ext>.samples,...`, scripts/data/prepare_timit.py:35), one (loader, transform,
batcher) triple per modality, `collate` to padded batches sorted longest first.  Host-side plumbing."""
import csv
import os
from typing import List, Tuple

from torch.utils.data import Dataset


class BaseDataset(Dataset):
    def __init__(self, source: str, modalities: List[Tuple], sort: bool = True, root: str = None):
        self.source, self.modalities, self.sort = source, modalities, sort
        root = os.path.dirname(os.path.abspath(source)) if root is None else root
        with open(source, newline="") as f:
            rows = list(csv.DictReader(f))
        first = next(iter(rows[0])) if rows else "filename"
        self.examples = [r[first] if os.path.isabs(r[first]) else os.path.join(root, r[first]) for r in rows]
        self.rows = rows

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, idx):
        outs, metas = [], []
        for loader, transform, _ in self.modalities:
            x, meta = loader(self.examples[idx])
            outs.append(transform(x) if transform is not None else x)
            metas.append(meta)
        return tuple(outs), tuple(metas)

    def collate(self, batch):
        """[(outputs, metadata)] -> ((x, x_sl) per modality — unwrapped when there is one modality —, metadata).
        An empty shard of a data-parallel evaluation batch (length_samplers.py) collates to (None, None)."""
        if len(batch) == 0:
            return None, None
        if self.sort:
            batch = self.modalities[0][2].sort(batch, sort_modality_idx=0)
        outs, metas = zip(*batch)
        collated = [m[2].collate([o[i] for o in outs]) for i, m in enumerate(self.modalities)]
        metadata = [tuple(mt[i] for mt in metas) for i in range(len(self.modalities))]
        if len(self.modalities) == 1:
            return collated[0], metadata[0]
        return tuple(collated), tuple(metadata)
