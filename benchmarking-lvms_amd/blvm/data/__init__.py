from blvm.data.base_dataset import BaseDataset  # noqa: F401
