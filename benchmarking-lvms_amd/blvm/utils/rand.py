"""Global seeding helpers with the reference's names (blvm/utils/rand.py)."""
import random

import numpy as np
import torch


def set_seed(seed: int) -> None:
    """Seed python's `random`, numpy and torch (host and device generators)."""
    random.seed(seed)
    np.random.seed(seed % 2**32)
    torch.manual_seed(seed)


def get_random_seed() -> int:
    return random.randint(0, 2**32 - 1)
