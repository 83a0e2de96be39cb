"""Convolutional coders of the Clockwork-VAE with the reference's constructors, module tree and parameter names
(blvm/models/clockwork_vae/convolutional_coders.py:15-310).  The nn.Conv1d / nn.GroupNorm children hold parameters only;
every block runs as ONE K11/K6 autograd node on time-major channel-last tensors [L,B,C] (`forward_tm`), and the
reference's [B,C,T] entry points permute around it."""
import functools
from typing import List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from blvm import ops
from blvm._hip import BlvmHipError
from blvm.modules.convolutions import (ConvDepthwiseSeparable1d, ConvTransposeDepthwiseSeparable1d, require_channel_norm,
                                       require_relu)  # fmt: skip
from blvm.utils.convolutions import compute_conv_attributes_single


def _to_tm(x):
    if not x.is_cuda:
        raise BlvmHipError("blvm HIP kernels were handed a CPU tensor (no CPU fallback)")
    return x.permute(2, 0, 1).contiguous()


class TemporalResidual(nn.Module):
    """module(x) + x, x nearest-resampled in time when the module changed the length (convolutional_coders.py:15-26)."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module


class BlockSeparable(nn.Module):
    def __init__(self, channels_bottleneck, kernel_size, stride, dilation, activation_cls: nn.Module, transposed,
                 channels_factor: int = 4, bias: bool = False):  # fmt: skip
        super().__init__()
        sep_conv_obj = ConvTransposeDepthwiseSeparable1d if transposed else ConvDepthwiseSeparable1d
        channels_block = channels_factor * channels_bottleneck
        transform = nn.Sequential(
            nn.Conv1d(channels_bottleneck, channels_block, 1, bias=bias),
            activation_cls(),
            nn.GroupNorm(num_channels=channels_block, num_groups=channels_block),
            sep_conv_obj(
                in_channels=channels_block,
                out_channels=channels_bottleneck,
                kernel_size=kernel_size,
                stride=stride,
                dilation=dilation,
                normalization=nn.GroupNorm(num_channels=channels_block, num_groups=channels_block),
                activation=activation_cls(),
            ),
        )
        require_relu(transform[1], "BlockSeparable")
        require_channel_norm(transform[2], channels_block, "BlockSeparable")
        self.block = TemporalResidual(module=transform)
        self.stride, self.dilation, self.transposed, self.kernel_size = stride, dilation, transposed, kernel_size

    def forward_tm(self, x: torch.Tensor) -> torch.Tensor:
        conv1, _, norm1, sep = self.block.module
        C = conv1.in_channels
        b1 = conv1.bias if conv1.bias is not None else torch.zeros(conv1.out_channels, device=x.device)
        return ops.sep_block(x, self.stride, self.dilation, self.transposed, conv1.weight.view(-1, C), b1, norm1.weight, norm1.bias,
                             sep.depthwise_conv.weight, sep.depthwise_conv.bias, sep.norm.weight, sep.norm.bias,
                             sep.pointwise_conv.weight.view(C, -1), eps=norm1.eps)  # fmt: skip

    def forward(self, x):
        """Reference layout [B,C,T] (convolutional_coders.py:65-66)."""
        return self.forward_tm(_to_tm(x)).permute(1, 2, 0)


class BlockSimple(nn.Module):
    """Dense k-tap (transposed) convolution -> channel norm -> ReLU, plus the nearest-resampled input
    (convolutional_coders.py:69-91): `ops.dense_conv` (all taps in one K6 GEMM), K11 channel norm, the ReLU pass, K11 resample-add."""

    def __init__(self, channels, kernel_size, stride, dilation, activation_cls: nn.Module, transposed, bias: bool = False):
        super().__init__()
        conv_obj = nn.ConvTranspose1d if transposed else nn.Conv1d
        conv = conv_obj(channels, channels, kernel_size, stride=stride, dilation=dilation, bias=bias)
        norm = nn.GroupNorm(num_channels=channels, num_groups=channels)  # channel-wise normalisation
        nonl = activation_cls()
        require_relu(nonl, "BlockSimple")
        require_channel_norm(norm, channels, "BlockSimple")
        self.block = TemporalResidual(nn.Sequential(conv, norm, nonl))
        self.stride, self.dilation, self.transposed, self.kernel_size = stride, dilation, transposed, kernel_size

    def forward_tm(self, x: torch.Tensor) -> torch.Tensor:
        conv, norm, _ = self.block.module
        h = ops.dense_conv(x, conv.weight, conv.bias, self.stride, self.dilation, self.transposed)
        h = ops.scale_act(ops.chan_norm(h.contiguous(), norm.weight, norm.bias, norm.eps), 1.0, 0.0)
        return ops.resample_add(h, x)

    def forward(self, x):
        """Reference layout [B,C,T] (convolutional_coders.py:90-91)."""
        return self.forward_tm(_to_tm(x)).permute(1, 2, 0)


class _Levels(nn.Sequential):
    def forward_tm(self, x):
        for block in self:
            x = block.forward_tm(x)
        return x


def _per_level(value, num_levels: int, first_only: bool):
    """Normalise a `channels_in` / `channels_out` argument to one entry per level: None -> no projection anywhere; an int -> that
    width at the first level only (`first_only`: inputs enter at the bottom) or at every level; a list is taken as given."""
    if value is None:
        return [None] * num_levels
    if isinstance(value, int):
        return [value] + [None] * (num_levels - 1) if first_only else [value] * num_levels
    return list(value)


def _level_plan(level_stride: int, stride_per_block: int, num_blocks: int, dilation_factor: int, level: int):
    """(stride, dilation) of the blocks of one level: the level's stride is spent `stride_per_block` at a time from the first block
    on, the remaining blocks have stride 1; block b is dilated by dilation_factor**b (convolutional_coders.py:185-191)."""
    plan, left = [], level_stride
    for b in range(num_blocks):
        take = stride_per_block if left >= stride_per_block else 1
        if take == 1 and left != 1:
            raise ValueError(f"remaining_stride={left} is not 1 at l={level}, b={b}.")
        left //= take
        plan.append((take, dilation_factor**b))
    return plan


def _receptive_field(plan, kernel_size: int, stride_in: int = 1, rf_in: int = 1):
    """Accumulated (stride, receptive field) behind a sequence of (stride, dilation) convolutions."""
    for stride, dilation in plan:
        _, stride_in, rf_in, _ = compute_conv_attributes_single(i=1, k=kernel_size, p=0, s=stride, d=dilation, s_in=stride_in, r_in=rf_in)
    return stride_in, rf_in


class ConvCoder1d(nn.Module):
    def __init__(self, strides: List[int], channels: int = 128, kernel_size: Union[int, List[int]] = 5, stride_per_block: int = 2,
                 dilation_factor: int = 1, num_blocks: int = 8, channels_in: Optional[Union[int, List[Union[None, int]]]] = None,
                 channels_out: Optional[Union[int, List[Union[None, int]]]] = None, transposed: bool = False,
                 block_type: str = "BlockSeparable", activation: nn.Module = nn.PReLU):  # fmt: skip
        """Same arguments, attributes, state_dict keys and parameter creation order as the reference's coder
        (convolutional_coders.py:94-231).  Built in two passes: the (stride, dilation) plan and receptive fields of every level
        are plain arithmetic (`_level_plan`, `_receptive_field`); the modules follow from the plans."""
        super().__init__()
        blocks_by_name = {"BlockSeparable": BlockSeparable, "BlockSimple": BlockSimple}
        if block_type not in blocks_by_name:
            raise ValueError(f"Unknown {block_type=}.")
        assert all(stride_per_block**num_blocks >= s for s in strides), f"Not enough blocks per level for {strides=}"
        self.strides, self.channels, self.kernel_size, self.num_blocks = strides, channels, kernel_size, num_blocks
        self.transposed, self.stride_per_block, self.block_type, self.activation = transposed, stride_per_block, block_type, activation
        self.num_levels = len(strides)
        self.overall_strides = np.cumprod(strides)
        self.overall_stride = self.overall_strides[-1]
        self.channels_in = _per_level(channels_in, self.num_levels, first_only=True)
        self.channels_out = _per_level(channels_out, self.num_levels, first_only=False)
        self.e_size = [channels if c is None else c for c in self.channels_out]

        # pass 1: arithmetic
        plans = [_level_plan(s, stride_per_block, num_blocks, dilation_factor, l) for l, s in enumerate(strides)]
        self.receptive_fields = [_receptive_field(plan, kernel_size)[1] for plan in plans]
        self.overall_receptive_fields, seen = [], (1, 1)
        for plan in plans:  # through all levels below as well
            seen = _receptive_field(plan, kernel_size, *seen)
            self.overall_receptive_fields.append(seen[1])
        self.overall_receptive_field = self.overall_receptive_fields[-1]

        # pass 2: modules, level by level (the reference creates a level's blocks, then its out projection, then its in projection:
        # seeded initialisation and state_dict order depend on it)
        self.levels, self.out_projs, self.in_projs = nn.ModuleList(), nn.ModuleDict(), nn.ModuleDict()
        for l, plan in enumerate(plans):
            blocks = [blocks_by_name[block_type](channels, kernel_size, stride, dilation, activation, transposed, bias=True) for stride, dilation in plan]
            self.levels.append(_Levels(*(reversed(blocks) if transposed else blocks)))  # a transposed coder mirrors the stride order
            for width, table, make in ((self.channels_out[l], self.out_projs, lambda c: nn.Conv1d(channels, c, 1)),
                                       (self.channels_in[l], self.in_projs, lambda c: nn.Conv1d(c, channels, 1))):  # fmt: skip
                if width is not None:
                    table[str(l)] = nn.Sequential(make(width), activation())
                    require_relu(table[str(l)][1], "ConvCoder1d.out_projs" if table is self.out_projs else "ConvCoder1d.in_projs")

    @property
    def device(self):
        return self.levels[0][0].block.module[0].weight.device

    def pad_level_tm(self, hidden: torch.Tensor, pad_left: int, pad_right: int):
        """Zero-pad (encoder, before the level) or crop (transposed coder, after the level) the TIME axis = dim 0
        (convolutional_coders.py:252-275)."""
        if not pad_left and not pad_right:
            return hidden
        if self.transposed:
            return hidden[pad_left : hidden.shape[0] - pad_right]
        return torch.nn.functional.pad(hidden, [0, 0, 0, 0, pad_left, pad_right])

    @staticmethod
    def _proj_tm(seq: nn.Sequential, x: torch.Tensor):
        L, B, C = x.shape
        w = seq[0].weight
        return ops.linear(x.reshape(L * B, C), w.view(w.shape[0], C), seq[0].bias, act=ops.ACT_RELU).view(L, B, -1)

    def forward_level_tm(self, hidden: torch.Tensor, level: int, pad_left: int = 0, pad_right: int = 0):
        """hidden [L,B,C] -> (hidden', encoding), both time-major (convolutional_coders.py:277-291)."""
        if str(level) in self.in_projs:
            hidden = self._proj_tm(self.in_projs[str(level)], hidden)
        if not self.transposed:
            hidden = self.pad_level_tm(hidden, pad_left, pad_right)
        hidden = self.levels[level].forward_tm(hidden)
        if self.transposed:
            hidden = self.pad_level_tm(hidden, pad_left, pad_right)
        encoding = self._proj_tm(self.out_projs[str(level)], hidden) if str(level) in self.out_projs else hidden
        return hidden, encoding

    def forward_tm(self, hidden: torch.Tensor, pad_left: List[int] = None, pad_right: List[int] = None):
        pad_left = [0] * self.num_levels if pad_left is None else pad_left
        pad_right = [0] * self.num_levels if pad_right is None else pad_right
        encodings = []
        for level in range(self.num_levels):
            hidden, encoding = self.forward_level_tm(hidden, level, pad_left[level], pad_right[level])
            encodings.append(encoding)
        return encodings

    # ---- reference layout [B,C,T] ------------------------------------------------------------------------------------
    def forward_level(self, hidden, level: int, pad_left: int = 0, pad_right: int = 0):
        h, e = self.forward_level_tm(_to_tm(hidden), level, pad_left, pad_right)
        return h.permute(1, 2, 0), e.permute(1, 2, 0)

    def forward(self, hidden, pad_left: List[int] = None, pad_right: List[int] = None):
        return [e.permute(1, 2, 0) for e in self.forward_tm(_to_tm(hidden), pad_left, pad_right)]

    def __getitem__(self, level: int):
        return functools.partial(self.forward_level, level=level)

    def __len__(self):
        return self.num_levels
