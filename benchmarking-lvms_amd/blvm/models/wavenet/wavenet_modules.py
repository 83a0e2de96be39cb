"""WaveNet building blocks with the reference's names, constructor arguments and parameter layout
(blvm/models/wavenet/wavenet_modules.py).  The modules own `nn.Conv1d` / `nn.Linear` parameters (same state_dict keys
and seeded initialisation as the reference); their arithmetic runs through K10/K6 on TIME-MAJOR channel-last tensors
[L, B, C] — the reference's [B, C, L] is transposed once at the model boundary.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from blvm import ops
from blvm.modules.activations import GatedTanhUnit


class CausalConv1d(nn.Module):
    """y[t] depends on x[:t] only: the last input is dropped before a kernel-size-k convolution (:14-50)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, activation: Optional[nn.Module] = None, **kwargs):
        super().__init__()
        if kernel_size not in (1, 2) or kwargs.get("groups", 1) != 1 or activation is not None:
            raise NotImplementedError("libblvm_hip: CausalConv1d is built for kernel_size in {1, 2}, groups=1, no activation")
        self.kernel_size = kernel_size
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, **kwargs)
        self.activation = None

    def init_weights_for_test(self):
        self.conv.weight.data.fill_(1)
        self.conv.bias.data.fill_(0)

    def forward_tm(self, x: torch.Tensor, pad_causal: bool = True):
        """x [L,B,C_in] time-major -> [L - pad_causal - (k-1), B, C_out]."""
        if pad_causal:
            x = x[:-1]
        if self.kernel_size == 1:
            L, B, C = x.shape
            return ops.linear(x.reshape(L * B, C), self.conv.weight.view(self.conv.out_channels, C), self.conv.bias).view(L, B, -1)
        return ops.conv1d_k2(x.contiguous(), self.conv.weight, self.conv.bias, 1)

    def forward(self, x: torch.Tensor, pad_causal: bool = True):
        """Reference layout: x [B,C,L] -> [B,C_out,L']."""
        return self.forward_tm(x.permute(2, 0, 1).contiguous(), pad_causal).permute(1, 2, 0)


class Conv1dResidualGLU(nn.Module):
    """Parameter container of one gated residual block (:53-117); executed by `ResidualStack` through K10."""

    def __init__(self, res_channels: int, skip_channels: Optional[int] = None, gate_channels: Optional[int] = None,
                 kernel_size: int = 2, dilation: int = 1, bias: bool = True, activation: nn.Module = GatedTanhUnit):  # fmt: skip
        super().__init__()
        skip_channels = res_channels if skip_channels is None else skip_channels
        gate_channels = 2 * res_channels if gate_channels is None else gate_channels
        if kernel_size != 2 or gate_channels != 2 * res_channels or not bias or activation is not GatedTanhUnit:
            raise NotImplementedError("libblvm_hip: residual blocks are built for kernel_size=2, gate_channels=2*res_channels")
        self.res_channels, self.skip_channels, self.gate_channels = res_channels, skip_channels, gate_channels
        self.kernel_size, self.dilation, self.bias = kernel_size, dilation, bias
        self.inv_std = math.sqrt(0.5)
        self.conv = nn.Conv1d(res_channels, gate_channels, kernel_size=kernel_size, dilation=dilation)
        self.conv1x1rs = nn.Conv1d(gate_channels // 2, res_channels + skip_channels, kernel_size=1, bias=bias)
        self.activation = activation(dim=1)

    def kernel_params(self):
        c = self.conv1x1rs
        return (self.conv.weight, self.conv.bias, c.weight.view(c.out_channels, c.in_channels), c.bias)


class ResidualStack(nn.Module):
    def __init__(self, n_layers: int, n_stacks: int, res_channels: int, skip_channels: Optional[int] = None,
                 gate_channels: Optional[int] = None, kernel_size: int = 2, base_dilation: int = 2, in_channels: int = None,
                 activation: nn.Module = GatedTanhUnit):  # fmt: skip
        super().__init__()
        in_channels = res_channels if in_channels is None else in_channels
        self.n_layers, self.n_stacks, self.res_channels = n_layers, n_stacks, res_channels
        self.skip_channels, self.gate_channels = skip_channels, gate_channels
        self.kernel_size, self.base_dilation, self.in_channels = kernel_size, base_dilation, in_channels
        self.dilations = self.build_dilations(n_layers, n_stacks, base_dilation)
        self.receptive_fields = self.compute_receptive_field(n_layers, n_stacks, kernel_size, base_dilation)
        self.receptive_field = self.receptive_fields[-1]
        # the reference always builds and applies this 1x1 convolution (in_channels is defaulted before the None test,
        # wavenet_modules.py:145,161-162 — SURVEY quirk 11)
        self.in_transform = nn.Conv1d(in_channels, res_channels, kernel_size=1)
        self.res_blocks = nn.ModuleList(
            Conv1dResidualGLU(res_channels=res_channels, skip_channels=skip_channels, gate_channels=gate_channels,
                              kernel_size=kernel_size, dilation=d, activation=activation) for d in self.dilations
        )  # fmt: skip

    @staticmethod
    def build_dilations(n_layers: int, n_stacks: int, base_dilation: int):
        if base_dilation > 1:
            return [1, *[base_dilation * 2**i for i in range(0, n_layers - 1)]] * n_stacks
        return [1] * n_layers * n_stacks

    @staticmethod
    def compute_receptive_field(n_layers: int, n_stacks: int, kernel_size: int, base_dilation: int):
        """Receptive field after every block: r_i = r_{i-1} + (k-1) d_i with stride 1 (utils/convolutions.py:83-210)."""
        dilations = [1, *[base_dilation * 2**i for i in range(0, n_layers - 1)]] * n_stacks
        r, out = 1, []
        for d in dilations:
            r += (kernel_size - 1) * d
            out.append(r)
        return out

    def forward_tm(self, x: torch.Tensor, skip_size: int, groups=None):
        """x [L,B,C_in] -> sum of the blocks' skip outputs [skip_size,B,S] (the reference returns the list and sums it
        in WaveNet.forward, wavenet.py:197-198).  `groups` (one entry per block: output index or -1) instead returns a
        tuple of partial sums — STCN reads only the last skip of every stack (stcn.py:299)."""
        L, B, C = x.shape
        t = self.in_transform
        o = ops.linear(x.reshape(L * B, C), t.weight.view(t.out_channels, C), t.bias).view(L, B, -1)
        blk = self.res_blocks[0]
        return ops.wavenet_stack(o, [b.kernel_params() for b in self.res_blocks], self.dilations, skip_size, blk.inv_std,
                                 blk.skip_channels, groups=groups)  # fmt: skip


class PointwiseTransform(nn.Module):
    """ReLU -> Linear -> ReLU on channel-last tensors (:214-239)."""

    def __init__(self, in_channels: int, out_channels: int, activation: nn.Module = nn.ReLU):
        super().__init__()
        if activation is not nn.ReLU:
            raise NotImplementedError("libblvm_hip: PointwiseTransform is built for ReLU")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.act1 = activation()
        self.linear = nn.Linear(in_channels, out_channels)
        self.act2 = activation()

    def forward_rows(self, x2d: torch.Tensor, scale: float = 1.0):
        """act2(Linear(act1(scale * x))) on [rows, C]."""
        return ops.linear(ops.scale_act(x2d, scale, 0.0), self.linear.weight, self.linear.bias, ops.ACT_RELU, 0.0)
