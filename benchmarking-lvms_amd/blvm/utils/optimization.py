"""Optimizer introspection with the reference's names (blvm/utils/optimization.py)."""
from typing import Dict

import torch


def get_learning_rates_dict(optimizer: torch.optim.Optimizer) -> Dict[str, float]:
    """{"lr": value} for one parameter group, {"lr_i": value} per group otherwise."""
    groups = optimizer.param_groups
    if len(groups) == 1:
        return {"lr": groups[0]["lr"]}
    return {f"lr_{i}": float(g["lr"]) for i, g in enumerate(groups)}
