"""Checkpoint / resume of a whole run with the reference's names and file layout (blvm/training/restore.py:16-80):
`model.save(directory)` (three files, base_model.py) plus `checkpoint.pt` with epoch / optimizer / scheduler state."""
import os
from typing import Optional

import torch

from blvm.models.base_model import BaseModel, load_model

CHECKPOINT_STR = "checkpoint.pt"


def save_run(directory: str, model: torch.nn.Module = None, optimizer=None, lr_scheduler=None, scaler=None, tracker=None):
    os.makedirs(directory, exist_ok=True)
    checkpoint = dict(
        epoch=getattr(tracker, "epoch", None),
        optimizer_state_dict=optimizer.state_dict() if optimizer is not None else None,
        lr_scheduler_state_dict=lr_scheduler.state_dict() if lr_scheduler is not None else None,
        scaler_state_dict=scaler.state_dict() if scaler is not None else None,
    )
    model.save(directory)
    torch.save(checkpoint, os.path.join(directory, CHECKPOINT_STR))


def load_run(directory: str, model: torch.nn.Module = None, optimizer=None, lr_scheduler=None, scaler=None,
             device: Optional[torch.device] = None, raise_errors: bool = True):  # fmt: skip
    """Returns (model, checkpoint dict); restores optimizer / scheduler / scaler state in place when given."""
    if isinstance(model, BaseModel):
        model.load_state_dict(torch.load(os.path.join(directory, "model_state_dict.pt"), map_location=device, weights_only=True))
    elif model is None:
        model = load_model(directory, device=device)
    path = os.path.join(directory, CHECKPOINT_STR)
    if not os.path.exists(path):
        if raise_errors:
            raise FileNotFoundError(path)
        return model, None
    checkpoint = torch.load(path, map_location=device, weights_only=True)  # written by save_run above
    for obj, key in ((optimizer, "optimizer_state_dict"), (lr_scheduler, "lr_scheduler_state_dict"), (scaler, "scaler_state_dict")):
        if obj is not None and checkpoint.get(key) is not None:
            obj.load_state_dict(checkpoint[key])
    return model, checkpoint
