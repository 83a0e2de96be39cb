"""Gated tanh unit (blvm/modules/activations.py:5-13); inside WaveNet blocks it is fused into K10."""
import torch
import torch.nn as nn


class GatedTanhUnit(nn.Module):
    def __init__(self, dim: int = -1) -> None:
        super().__init__()
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a, b = x.chunk(2, self.dim)
        return torch.tanh(a) * torch.sigmoid(b)
