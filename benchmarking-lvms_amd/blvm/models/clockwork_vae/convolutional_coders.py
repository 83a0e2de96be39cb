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
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("BlockSimple (dense k-tap conv blocks) has no HIP kernel yet; CWVAEAudio uses BlockSeparable")


class _Levels(nn.Sequential):
    def forward_tm(self, x):
        for block in self:
            x = block.forward_tm(x)
        return x


class ConvCoder1d(nn.Module):
    def __init__(self, strides: List[int], channels: int = 128, kernel_size: Union[int, List[int]] = 5, stride_per_block: int = 2,
                 dilation_factor: int = 1, num_blocks: int = 8, channels_in: Optional[Union[int, List[Union[None, int]]]] = None,
                 channels_out: Optional[Union[int, List[Union[None, int]]]] = None, transposed: bool = False,
                 block_type: str = "BlockSeparable", activation: nn.Module = nn.PReLU):  # fmt: skip
        """Same arguments as the reference (convolutional_coders.py:94-124)."""
        super().__init__()
        if block_type not in ["BlockSeparable", "BlockSimple"]:
            raise ValueError(f"Unknown {block_type=}.")
        num_levels = len(strides)
        overall_strides = np.cumprod(strides)
        assert all(stride_per_block**num_blocks >= s for s in strides), f"Not enough blocks per level for {strides=}"

        self.strides, self.channels, self.kernel_size, self.num_blocks = strides, channels, kernel_size, num_blocks
        self.transposed, self.stride_per_block, self.block_type, self.activation = transposed, stride_per_block, block_type, activation
        self.num_levels, self.overall_strides, self.overall_stride = num_levels, overall_strides, overall_strides[-1]

        if channels_in is None:
            self.channels_in = [None] * num_levels
        elif isinstance(channels_in, int):
            self.channels_in = [channels_in] + [None] * (num_levels - 1)
        else:
            self.channels_in = channels_in
        if channels_out is None:
            self.channels_out = [None] * num_levels
        elif isinstance(channels_out, int):
            self.channels_out = [channels_out] * num_levels
        else:
            self.channels_out = channels_out
        self.e_size = [c if c is not None else self.channels for c in self.channels_out]

        block_cls = {"BlockSeparable": BlockSeparable, "BlockSimple": BlockSimple}[block_type]
        self.overall_receptive_fields, self.receptive_fields = [], []
        self.levels = nn.ModuleList()
        self.out_projs = nn.ModuleDict()
        self.in_projs = nn.ModuleDict()

        overall_stride_in, overall_rf_in = 1, 1
        for l in range(num_levels):
            remaining_stride = self.strides[l]
            stride_in, rf_in = 1, 1
            blocks = []
            for b in range(num_blocks):
                dilation = dilation_factor**b
                if remaining_stride >= self.stride_per_block:
                    stride = self.stride_per_block
                    remaining_stride = remaining_stride // self.stride_per_block
                else:
                    if remaining_stride != 1:
                        raise ValueError(f"{remaining_stride=} is not 1 at {l=}, {b=}.")
                    stride = 1
                blocks.append(block_cls(channels, kernel_size, stride, dilation, activation, transposed, bias=True))
                _, overall_stride_in, overall_rf_in, _ = compute_conv_attributes_single(
                    i=1, k=kernel_size, p=0, s=stride, d=dilation, s_in=overall_stride_in, r_in=overall_rf_in)  # fmt: skip
                _, stride_in, rf_in, _ = compute_conv_attributes_single(
                    i=1, k=kernel_size, p=0, s=stride, d=dilation, s_in=stride_in, r_in=rf_in)  # fmt: skip
            self.overall_receptive_fields.append(overall_rf_in)
            self.receptive_fields.append(rf_in)
            if transposed:  # mirrored stride order (convolutional_coders.py:227-231)
                blocks = blocks[::-1]
            self.levels.append(_Levels(*blocks))
            # parameter creation order of the reference: out projection, then in projection
            if self.channels_out[l] is not None:
                self.out_projs[str(l)] = nn.Sequential(nn.Conv1d(channels, self.channels_out[l], 1), activation())
                require_relu(self.out_projs[str(l)][1], "ConvCoder1d.out_projs")
            if self.channels_in[l] is not None:
                self.in_projs[str(l)] = nn.Sequential(nn.Conv1d(self.channels_in[l], channels, 1), activation())
                require_relu(self.in_projs[str(l)][1], "ConvCoder1d.in_projs")
        self.overall_receptive_field = self.overall_receptive_fields[-1]

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
