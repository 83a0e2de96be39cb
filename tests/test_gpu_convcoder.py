"""GPU (MI355X): K11 — the Clockwork-VAE convolutional-coder kernels through the C ABI against the torch CPU operators
the reference builds its coders from (nn.GroupNorm(groups=C), depthwise nn.Conv1d / nn.ConvTranspose1d, F.interpolate
'nearest'; blvm/models/clockwork_vae/convolutional_coders.py:15-66, blvm/modules/convolutions.py:6-104), evaluated in
float64.  Tolerance: relative L2 <= 2e-5 on outputs and gradients (fp32 kernels, fp64 statistics)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from blvm import _hip, ops
from blvm.models.clockwork_vae.convolutional_coders import BlockSeparable, BlockSimple, ConvCoder1d

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5


@pytest.fixture(scope="module", autouse=True)
def _require_hip():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert _hip.load().blvm_device_ok() == 1


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def tm(x_bct):  # [B,C,T] -> [T,B,C]
    return x_bct.permute(2, 0, 1).contiguous()


def bct(x_tm):
    return x_tm.permute(1, 2, 0)


@pytest.mark.parametrize("L,B,C", [(37, 3, 8), (1000, 2, 64), (5, 1, 4), (4097, 2, 12)])
def test_chan_norm_matches_groupnorm(L, B, C):
    g = torch.Generator().manual_seed(L + B + C)
    x = (torch.randn(B, C, L, generator=g) * 3 + 1.5).double().requires_grad_()
    gn = nn.GroupNorm(C, C).double()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
    dy = torch.randn(B, C, L, generator=g).double()
    y = gn(x)
    y.backward(dy)

    xd = tm(x.detach().float()).to(DEV).requires_grad_()
    w, b = gn.weight.detach().float().to(DEV).requires_grad_(), gn.bias.detach().float().to(DEV).requires_grad_()
    yd = ops.chan_norm(xd, w, b, gn.eps)
    yd.backward(tm(dy.float()).to(DEV))
    assert rel_l2(bct(yd), y) < TOL
    assert rel_l2(bct(xd.grad), x.grad) < 5 * TOL
    assert rel_l2(w.grad, gn.weight.grad) < TOL and rel_l2(b.grad, gn.bias.grad) < TOL


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("stride,dilation,relu", [(1, 1, False), (2, 1, True), (4, 1, True), (2, 2, False), (3, 1, True)])
@pytest.mark.parametrize("L,B,C,k", [(50, 2, 8, 5), (333, 3, 16, 5), (9, 1, 4, 3)])
def test_depthwise_conv_matches_torch(transposed, stride, dilation, relu, L, B, C, k):
    g = torch.Generator().manual_seed(L * 3 + stride + 7 * dilation + transposed)
    x = torch.randn(B, C, L, generator=g).double().requires_grad_()
    w = torch.randn(C, 1, k, generator=g).double().requires_grad_()
    b = torch.randn(C, generator=g).double().requires_grad_()
    fn = F.conv_transpose1d if transposed else F.conv1d
    y = fn(x, w, b, stride=stride, dilation=dilation, groups=C)
    y = torch.relu(y) if relu else y
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)

    xd = tm(x.detach().float()).to(DEV).requires_grad_()
    wd, bd = w.detach().float().to(DEV).requires_grad_(), b.detach().float().to(DEV).requires_grad_()
    yd = ops.dwconv(xd, wd, bd, stride, dilation, transposed, relu)
    assert yd.shape == (y.shape[2], B, C)
    yd.backward(tm(dy.float()).to(DEV))
    assert rel_l2(bct(yd), y) < TOL
    assert rel_l2(bct(xd.grad), x.grad) < TOL
    assert rel_l2(wd.grad, w.grad) < TOL and rel_l2(bd.grad, b.grad) < TOL


def test_depthwise_conv_rejects_short_input():
    x = torch.randn(3, 1, 4, device=DEV)
    with pytest.raises(_hip.BlvmHipError):
        ops.dwconv(x, torch.randn(4, 1, 5, device=DEV), torch.zeros(4, device=DEV))


@pytest.mark.parametrize("Lx,Ly", [(100, 100), (100, 48), (48, 99), (7, 31), (1000, 251), (3, 3 * 16 + 4)])
def test_resample_add_matches_interpolate_nearest(Lx, Ly):
    g = torch.Generator().manual_seed(Lx + Ly)
    B, C = 2, 8
    x = torch.randn(B, C, Lx, generator=g).requires_grad_()
    y = torch.randn(B, C, Ly, generator=g).requires_grad_()
    out = y + (F.interpolate(x, size=Ly, mode="nearest") if Lx != Ly else x)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    xd, yd = tm(x.detach()).to(DEV).requires_grad_(), tm(y.detach()).to(DEV).requires_grad_()
    od = ops.resample_add(yd, xd)
    od.backward(tm(dout).to(DEV))
    assert torch.equal(bct(od).cpu(), out.detach())  # index arithmetic identical to torch's -> bit-exact
    assert torch.equal(bct(yd.grad).cpu(), y.grad)
    assert rel_l2(bct(xd.grad), x.grad) < 1e-6  # atomics: order of the (<= ceil(Ly/Lx)) adds is free


def _torch_block(block: BlockSeparable, x):
    """The reference's forward of one block, from the parameter-holding torch modules, in float64 on the CPU."""
    conv1, act, norm1, sep = block.block.module
    h = norm1(act(conv1(x)))
    h = sep.pointwise_conv(sep.norm(sep.activation(sep.depthwise_conv(h))))
    return h + (x if h.shape[-1] == x.shape[-1] else F.interpolate(x, size=h.shape[-1], mode="nearest"))


def _randomise_norms(module, g):
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, nn.GroupNorm):
                m.weight.copy_(1 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.3 * torch.randn(m.bias.shape, generator=g))


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("stride,C,B,L", [(1, 24, 3, 61), (2, 24, 3, 61), (4, 24, 3, 61), (2, 192, 2, 777), (1, 192, 2, 300)])
def test_separable_block_matches_torch(transposed, stride, C, B, L):
    """Last two cases: BASELINE C4 widths (192 -> 768 channels)."""
    torch.manual_seed(stride + 10 * transposed)
    g = torch.Generator().manual_seed(1)
    block = BlockSeparable(C, 5, stride, 1, nn.ReLU, transposed, bias=True)
    _randomise_norms(block, g)
    ref = BlockSeparable(C, 5, stride, 1, nn.ReLU, transposed, bias=True).double()
    ref.load_state_dict(block.state_dict())
    x = torch.randn(B, C, L, generator=g)
    xr = x.double().requires_grad_()
    yr = _torch_block(ref, xr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    block = block.to(DEV)
    xd = x.to(DEV).requires_grad_()
    yd = block(xd)  # reference layout entry point [B,C,T]
    assert yd.shape == yr.shape
    yd.backward(dy.to(DEV))
    assert rel_l2(yd, yr) < TOL
    assert rel_l2(xd.grad, xr.grad) < 5 * TOL
    pr = dict(ref.named_parameters())
    for n, p in block.named_parameters():
        assert p.grad is not None, n
        assert rel_l2(p.grad, pr[n].grad) < 3e-4, n  # fp32 GEMM weight gradients over L*B rows vs float64


def _torch_simple_block(block: BlockSimple, x):
    """`BlockSimple.forward` (convolutional_coders.py:69-91) from the parameter-holding torch modules, in float64 on the CPU."""
    h = block.block.module(x)  # conv -> GroupNorm(groups = channels) -> ReLU
    return h + (x if h.shape[-1] == x.shape[-1] else F.interpolate(x, size=h.shape[-1], mode="nearest"))


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("stride,dilation,C,B,L,k", [(1, 1, 16, 3, 61, 5), (2, 1, 16, 3, 61, 5), (4, 1, 24, 2, 203, 5), (2, 2, 8, 2, 50, 3), (3, 1, 8, 1, 40, 4)])
def test_simple_block_matches_torch(transposed, stride, dilation, C, B, L, k):
    """`BlockSimple` (dense k-tap Conv1d / ConvTranspose1d -> channel norm -> ReLU, + resampled input): all taps as one GEMM and
    strided tap sums, against the torch operators in float64 — output, input gradient and every parameter gradient."""
    torch.manual_seed(stride + 10 * transposed + dilation)
    g = torch.Generator().manual_seed(7)
    block = BlockSimple(C, k, stride, dilation, nn.ReLU, transposed, bias=True)
    _randomise_norms(block, g)
    ref = BlockSimple(C, k, stride, dilation, nn.ReLU, transposed, bias=True).double()
    ref.load_state_dict(block.state_dict())
    x = torch.randn(B, C, L, generator=g)
    xr = x.double().requires_grad_()
    yr = _torch_simple_block(ref, xr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    block = block.to(DEV)
    xd = x.to(DEV).requires_grad_()
    yd = block(xd)  # reference layout entry point [B,C,T]
    assert yd.shape == yr.shape
    yd.backward(dy.to(DEV))
    assert rel_l2(yd, yr) < TOL
    assert rel_l2(xd.grad, xr.grad) < 5 * TOL
    pr = dict(ref.named_parameters())
    for n, p in block.named_parameters():
        assert p.grad is not None, n
        if n == "block.module.0.bias":  # a bias in front of a per-channel norm over time: its gradient is exactly zero (rounding noise on both sides)
            assert float(p.grad.abs().max()) < 1e-4 and float(pr[n].grad.abs().max()) < 1e-10
        else:
            assert rel_l2(p.grad, pr[n].grad) < 3e-4, n


def test_conv_coder_of_simple_blocks_runs_and_keeps_the_reference_layout():
    """ConvCoder1d(block_type="BlockSimple"): same state_dict keys as the module tree the reference builds (block.module.{0,1}),
    levels strided as planned, encodings against the torch modules."""
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(3)
    kw = dict(strides=[4, 2], channels=8, kernel_size=5, num_blocks=2, stride_per_block=2, transposed=False, activation=nn.ReLU,
              block_type="BlockSimple", channels_in=3)
    coder = ConvCoder1d(**kw)
    _randomise_norms(coder, g)
    keys = [k for k in coder.state_dict() if k.startswith("levels.0.0.")]
    assert keys == [f"levels.0.0.block.module.{i}.{p}" for i in (0, 1) for p in ("weight", "bias")]
    ref = ConvCoder1d(**kw).double()
    ref.load_state_dict(coder.state_dict())
    x = torch.randn(2, 3, 203, generator=g)
    pads = [7, 3]
    h, encs_r = x.double(), []
    for l in range(2):
        if str(l) in ref.in_projs:
            h = ref.in_projs[str(l)](h)
        h = F.pad(h, [0, pads[l]])
        for blk in ref.levels[l]:
            h = _torch_simple_block(blk, h)
        encs_r.append(ref.out_projs[str(l)](h) if str(l) in ref.out_projs else h)
    encs = coder.to(DEV)(x.to(DEV), pad_right=pads)
    for e, er in zip(encs, encs_r):
        assert e.shape == er.shape and rel_l2(e, er) < TOL


@pytest.mark.parametrize("transposed", [False, True])
def test_conv_coder_levels_match_torch(transposed):
    """Two-level coder with projections and same-padding / cropping, against the torch modules it holds."""
    torch.manual_seed(3 + transposed)
    g = torch.Generator().manual_seed(2)
    C, B = 16, 2
    kw = dict(strides=[4, 2], channels=C, kernel_size=5, num_blocks=2, stride_per_block=2, transposed=transposed, activation=nn.ReLU)
    kw.update(dict(channels_in=[6, 5], channels_out=[8, None]) if transposed else dict(channels_in=3))
    coder = ConvCoder1d(**kw)
    _randomise_norms(coder, g)
    ref = ConvCoder1d(**kw).double()
    ref.load_state_dict(coder.state_dict())
    coder = coder.to(DEV)

    def ref_level(hidden, level, pad):
        if str(level) in ref.in_projs:
            hidden = ref.in_projs[str(level)](hidden)
        if not transposed and pad:
            hidden = F.pad(hidden, [0, pad])
        for blk in ref.levels[level]:
            hidden = _torch_block(blk, hidden)
        if transposed and pad:
            hidden = F.pad(hidden, [0, -pad])
        enc = ref.out_projs[str(level)](hidden) if str(level) in ref.out_projs else hidden
        return hidden, enc

    if not transposed:
        x = torch.randn(B, 3, 203, generator=g)
        pads = [7, 3]
        xr = x.double().requires_grad_()
        h, encs_r = xr, []
        for l in range(2):
            h, e = ref_level(h, l, pads[l])
            encs_r.append(e)
        xd = x.to(DEV).requires_grad_()
        encs = coder(xd, pad_right=pads)
        dys = [torch.randn(e.shape, generator=g) for e in encs_r]
        sum((e * d.double()).sum() for e, d in zip(encs_r, dys)).backward()
        sum((e * d.to(DEV)).sum() for e, d in zip(encs, dys)).backward()
        for e, er in zip(encs, encs_r):
            assert e.shape == er.shape and rel_l2(e, er) < TOL
    else:
        x = torch.randn(B, 5, 9, generator=g)
        xr = x.double().requires_grad_()
        h, e1 = ref_level(xr, 1, 2)
        x0 = torch.randn(B, 6, e1.shape[-1], generator=g)
        _, e0 = ref_level(x0.double(), 0, 5)
        xd = x.to(DEV).requires_grad_()
        _, d1 = coder[1](xd, pad_right=2)
        _, d0 = coder[0](x0.to(DEV), pad_right=5)
        assert d1.shape == e1.shape and d0.shape == e0.shape
        assert rel_l2(d1, e1) < TOL and rel_l2(d0, e0) < TOL
        dy1, dy0 = torch.randn(e1.shape, generator=g), torch.randn(e0.shape, generator=g)
        ((e1 * dy1.double()).sum() + (e0 * dy0.double()).sum()).backward()
        ((d1 * dy1.to(DEV)).sum() + (d0 * dy0.to(DEV)).sum()).backward()
    assert rel_l2(xd.grad, xr.grad) < 1e-4
    pr = dict(ref.named_parameters())
    for n, p in coder.named_parameters():
        assert p.grad is not None, n
        assert rel_l2(p.grad, pr[n].grad) < 2e-4, n
