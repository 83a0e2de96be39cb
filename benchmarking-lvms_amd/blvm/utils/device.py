"""Device selection with the reference's names (blvm/utils/device.py): on this build the answer is always a HIP device."""
import os
from typing import Any

import torch


def get_device(idx: int = None) -> torch.device:
    """The HIP device of this process: LOCAL_RANK under torch.distributed.run, else `idx` or device 0."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the blvm hot path runs on gfx950 only (no CPU fallback)")
    if idx is None:
        idx = int(os.environ.get("LOCAL_RANK", "0"))
    return torch.device("cuda", idx)


def to_device_recursive(x: Any, device: torch.device):
    if isinstance(x, (torch.Tensor, torch.nn.Module)):
        return x.to(device)
    if isinstance(x, (list, tuple)):
        return type(x)(to_device_recursive(e, device) for e in x)
    return x
