"""Depthwise-separable 1-D convolutions with the reference's constructors and parameter layout
(blvm/modules/convolutions.py:6-104).  The nn.Conv1d / nn.GroupNorm children only HOLD parameters (same names, shapes and
init order as the reference, so seeds and state_dicts carry over); the arithmetic runs on K11/K6 (`blvm.ops`)."""
import torch
from torch import nn

from blvm import ops
from blvm._hip import BlvmHipError


def _single(v):
    return v if isinstance(v, tuple) else (v,)


def require_relu(activation, where: str):
    """The fused HIP kernels implement ReLU (what CWVAEAudio uses, clockwork_vae.py:470,484)."""
    if not isinstance(activation, nn.ReLU):
        raise NotImplementedError(f"{where}: only nn.ReLU is implemented on the HIP path, got {type(activation).__name__}")


def require_channel_norm(norm, channels: int, where: str):
    if not (isinstance(norm, nn.GroupNorm) and norm.num_groups == channels and norm.num_channels == channels and norm.affine):
        raise NotImplementedError(f"{where}: only channel-wise affine GroupNorm (num_groups == num_channels) is implemented")


class _DepthwiseSeparable1d(nn.Module):
    transposed = False

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 dilation: int = 1, normalization: nn.Module = None, activation: nn.Module = nn.ReLU()):  # fmt: skip
        super().__init__()
        self.stride, self.kernel_size = _single(stride), _single(kernel_size)
        self.padding, self.dilation = _single(padding), _single(dilation)
        if self.padding[0] != 0:
            raise NotImplementedError("depthwise-separable conv: callers pad (the reference's coders use padding=0)")
        conv_cls = nn.ConvTranspose1d if self.transposed else nn.Conv1d
        self.depthwise_conv = conv_cls(in_channels, in_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                                       groups=in_channels, bias=True)  # fmt: skip
        self.activation = activation
        self.norm = normalization
        self.pointwise_conv = nn.Conv1d(in_channels, out_channels, 1, bias=False)
        require_relu(activation, type(self).__name__)
        if normalization is not None:
            require_channel_norm(normalization, in_channels, type(self).__name__)

    def forward_tm(self, x: torch.Tensor) -> torch.Tensor:
        """x [L,B,C_in] (time-major, channel-last) -> [L',B,C_out]."""
        dw = self.depthwise_conv
        y = ops.dwconv(x, dw.weight, dw.bias, self.stride[0], self.dilation[0], self.transposed, relu=True)
        if self.norm is not None:
            y = ops.chan_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)
        L, B, C = y.shape
        pw = self.pointwise_conv.weight
        return ops.linear(y.view(L * B, C), pw.view(pw.shape[0], C), None).view(L, B, -1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Reference layout: x [B,C_in,T] -> [B,C_out,T'] (convolutions.py:43-54)."""
        if not x.is_cuda:
            raise BlvmHipError("blvm HIP kernels were handed a CPU tensor (no CPU fallback)")
        return self.forward_tm(x.permute(2, 0, 1).contiguous()).permute(1, 2, 0)


class ConvDepthwiseSeparable1d(_DepthwiseSeparable1d):
    transposed = False


class ConvTransposeDepthwiseSeparable1d(_DepthwiseSeparable1d):
    transposed = True
