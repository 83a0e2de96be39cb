"""Host-side shape logic the kernels must agree with (reference: blvm/utils/operations.py).

Pure index/shape arithmetic on (mostly host) tensors: frame stacking, masks, per-row reversal, sequence splitting.
"""
import math
from typing import List, Tuple, Union

import torch


def stack_tensor(x: torch.Tensor, stack_size: int, dim: int = -1) -> Tuple[torch.Tensor, int]:
    """Split `dim` into stacks of `stack_size` (right zero-padded) with a new right-most stack dimension
    (operations.py:14-32)."""
    if abs(dim) > x.ndim:
        raise ValueError(f"Got {dim=} which is out of range for x with shape {x.shape}")
    dim = dim if dim > 0 else x.ndim + dim
    padding = (-x.size(dim)) % stack_size
    if padding:
        pad = [0, 0] * (x.ndim - dim - 1) + [0, padding]
        x = torch.nn.functional.pad(x, pad)
    shape = [x.size(i) if i != dim else x.size(i) // stack_size for i in range(x.ndim)] + [stack_size]
    return x.reshape(*shape), padding


def unstack_tensor(x: torch.Tensor, stack_size: int, padding: int = 0, dim: int = -1) -> torch.Tensor:
    """Inverse of `stack_tensor` (operations.py:35-53)."""
    if abs(dim) > x.ndim:
        raise ValueError(f"Got {dim=} which is out of range for x with shape {x.shape}")
    dim = dim if dim > 0 else x.ndim + dim
    shape = [x.size(i) if i != (dim - 1) else x.size(i) * stack_size for i in range(x.ndim)]
    shape[-1] = -1
    x = x.reshape(*shape)
    if padding:
        x = x.narrow(dim - 1, 0, x.size(dim - 1) - padding)
    return x


def sequence_mask(seq_lens: Union[list, torch.Tensor], stride: int = 1, max_len: int = None,
                  dtype: torch.dtype = torch.bool, device: torch.device = None):  # fmt: skip
    """[N,T] mask with ones before `seq_lens` (operations.py:90-119)."""
    if isinstance(seq_lens, torch.Tensor):
        device = seq_lens.device if device is None else device
        seq_lens = seq_lens.to(device)
    else:
        seq_lens = torch.tensor(seq_lens, device=device, dtype=torch.int64)
    T = max_len or math.ceil(int(seq_lens.max()) / stride)
    return (torch.arange(T, device=device).unsqueeze(0) < seq_lens.unsqueeze(1)).to(dtype)


def reverse_sequences(x: torch.Tensor, x_sl: torch.Tensor, batch_first: bool = False):
    """Reverse [T,B,*] along time per row, leaving right padding in place (operations.py:56-87).  The index map
    is built once on the host side of the tensor's device; the gather itself is a single indexed copy."""
    if batch_first:
        x = x.transpose(0, 1)
    T = int(x_sl.max())
    sl = x_sl.to(x.device).unsqueeze(0)  # [1,B]
    t = torch.arange(T, device=x.device).unsqueeze(1)  # [T,1]
    idx = torch.where(t < sl, sl - 1 - t, t)  # [T,B]
    idx = idx.view(T, -1, *([1] * (x.ndim - 2))).expand(-1, -1, *x.shape[2:])
    out = torch.gather(x, 0, idx)
    return out.transpose(0, 1) if batch_first else out


def split_sequence(x: torch.Tensor, x_sl: torch.Tensor, length: int, overlap: int = 0, drop_inactive: bool = True,
                   mode: str = "consume") -> Tuple[List[torch.Tensor], List[torch.Tensor]]:  # fmt: skip
    """Split [B,T,*] into sub-sequences with optional overlap (operations.py:122-197)."""
    if mode == "consume":
        if overlap >= length:
            raise ValueError("`split_sequence` does not support `overlap >= length` in `consume` mode")
        n = math.ceil(x.size(1) / (length - overlap))
        start = [i * (length - overlap) for i in range(n)]
        stop = [s + length for s in start]
    elif mode == "extend":
        n = math.ceil(x.size(1) / length)
        start = [max(i * length - overlap, 0) for i in range(n)]
        stop = [(i + 1) * length for i in range(n)]
    else:
        raise ValueError(f"Unknown mode `{mode}`. Recognized options are `consume` and `extend`.")
    active = torch.ones(x.shape[0], dtype=torch.bool)
    xs, sls, i = [], [], 0
    while active.any():
        xs.append(x[active, start[i] : stop[i]] if drop_inactive else x[:, start[i] : stop[i]])
        new_active = x_sl > stop[i]
        sl = length * new_active + (x_sl - start[i]).clamp(0) * ~new_active
        sls.append(sl[active] if drop_inactive else sl)
        active = new_active
        i += 1
    return xs, sls


def detach(x):
    return x.detach() if isinstance(x, torch.Tensor) else x
