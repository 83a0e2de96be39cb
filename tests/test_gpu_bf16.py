"""The bf16-operand mode (`blvm_set_operand_dtype`, the reference's `--use_amp True` regime: `experiments/experiment_vrnn_audio.py:198,
219-230`, `benchmarks.txt:6-29`): bf16 operands / fp32 accumulation for the K6 GEMMs and the persistent recurrent chains.

Two kinds of checks: (1) the kernels compute exactly what the mode says — products of bf16-ROUNDED operands accumulated in fp32
(against float64 products of the rounded operands: tolerance = fp32 summation); (2) the mode's distance to the fp32 path on the
reference's golden inputs stays inside SURVEY A.4's budget: 1e-4 relative ELBO = 1.2e-3 nats/frame at random init."""
import os

import numpy as np
import pytest
import torch

import blvm_oracle as O
from blvm import _hip, ops
from blvm.models import SRNNAudio, VRNNAudio

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NATS_PER_FRAME_BUDGET = 1.2e-3


@pytest.fixture(autouse=True)
def _bf16_mode():
    assert torch.cuda.is_available() and _hip.load().blvm_device_ok() == 1
    _hip.set_operand_dtype("bf16")
    yield
    _hip.set_operand_dtype("f32")
    _hip.check_async()


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rb(t):  # what the kernels multiply: operands rounded to bf16 (nearest even)
    return t.to(torch.bfloat16).double()


def test_mode_switch_round_trip():
    assert _hip.get_operand_dtype() == "bf16"
    _hip.set_operand_dtype("f32")
    assert _hip.get_operand_dtype() == "f32"
    with pytest.raises(ValueError):
        _hip.set_operand_dtype("fp8")
    assert _hip.load().blvm_set_operand_dtype(7) != 0
    _hip.set_operand_dtype("bf16")


@pytest.mark.parametrize("op_a,op_b", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (130, 70, 50), (256, 1920, 768), (1000, 30, 30), (16, 256, 1003), (384, 192, 517)])
def test_gemm_bf16_operands_fp32_accumulate(op_a, op_b, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + op_a * 2 + op_b)
    A = torch.randn(*((K, M) if op_a else (M, K)), generator=g)
    Bm = torch.randn(*((K, N) if op_b else (N, K)), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (rb(A).t() if op_a else rb(A)) @ (rb(Bm) if op_b else rb(Bm).t()) + bias.double()
    ref = torch.where(ref > 0, ref, 0.01 * ref)
    C = torch.empty(M, N, device=DEV)
    ops.gemm(op_a, op_b, M, N, K, A.to(DEV), A.shape[1], Bm.to(DEV), Bm.shape[1], C, N, bias=bias.to(DEV), act=ops.ACT_LEAKY, slope=0.01)
    assert rel_l2(C, ref) < 2e-6
    # and it IS the reduced-precision product: the fp32 product of the unrounded operands is ~2^-9 away
    full = (A.t() if op_a else A).double() @ (Bm if op_b else Bm.t()).double() + bias.double()
    if K >= 50:
        assert rel_l2(C, torch.where(full > 0, full, 0.01 * full)) > 1e-4


@pytest.mark.parametrize("op_a,op_b,M,N,K,split", [(1, 1, 192, 768, 30000, 40), (1, 0, 384, 130, 20001, 24), (0, 1, 16385, 576, 192, 1), (1, 1, 96, 80, 4096, 16)])
def test_gemm_bf16_split_k_wide_tiles_accumulate(op_a, op_b, M, N, K, split):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(*((K, M) if op_a else (M, K)), generator=g)
    Bm = torch.randn(*((K, N) if op_b else (N, K)), generator=g)
    ref = (rb(A).t() if op_a else rb(A)) @ (rb(Bm) if op_b else rb(Bm).t())
    C = torch.full((M, N + 4), 2.0, device=DEV)
    ops.gemm(op_a, op_b, M, N, K, A.to(DEV), A.shape[1], Bm.to(DEV), Bm.shape[1], C, N + 4, accumulate=True, split_k=split)
    assert rel_l2(C[:, :N], ref + 2) < 3e-6
    assert torch.all(C[:, N:] == 2)


def test_wgrad_bias_gradient_stays_fp32_in_bf16_mode():
    g = torch.Generator().manual_seed(4)
    N, K, rows = 256, 192, 9000
    D, X = torch.randn(rows, N, generator=g), torch.randn(rows, K, generator=g)
    dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    Dd, Xd = D.to(DEV), X.to(DEV)
    _hip.check(_hip.load().blvm_wgrad_f32(N, K, rows, _hip.ptr(Dd), N, _hip.ptr(Xd), K, _hip.ptr(dW), K, _hip.ptr(db), 0, _hip.stream_ptr()), "wgrad")
    assert rel_l2(dW, rb(D).t() @ rb(X)) < 3e-6
    assert rel_l2(db, D.double().sum(0)) < 3e-6  # NOT the sum of the rounded operands


def _run(model, x, x_sl, eps, beta, fn):
    model.zero_grad()
    loss, metrics, out = model(x.to(DEV), x_sl, beta=beta, free_nats=fn, eps=eps.to(DEV))
    loss.backward()
    return loss, metrics, out


def test_vrnn_full_dims_bf16_delta_within_budget():
    """C2 dims on the golden inputs of `test_vrnn_full_dims_vs_reference_golden`: per-utterance ELBO / KL of the bf16-operand
    mode against the REFERENCE's fp32 values, in nats per frame, and the gradient direction."""
    g = np.load(os.path.join(GOLDEN, "vrnn_full.npz"))
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(DEV)
    x, x_sl = O.synth_batch(4, 1280, seed=0, ragged=True)
    torch.manual_seed(123)
    eps = torch.stack([torch.randn(4, 256) for _ in range(20)], 0)
    loss, metrics, out = _run(m, x, x_sl, eps, 1.0, 2.0)
    frames = x_sl.double()
    d_elbo = ((out.elbo.cpu().double() - T(g["elbo"]).double()).abs() / frames).max()
    d_kl = ((out.kl.cpu().double() - T(g["kl"]).double()).abs() / frames).max()
    d_elbo, d_kl = d_elbo.detach(), d_kl.detach()
    assert float(d_elbo) < NATS_PER_FRAME_BUDGET, float(d_elbo)
    assert float(d_kl) < NATS_PER_FRAME_BUDGET, float(d_kl)
    assert float(d_elbo) > 0.0  # the mode is on (the fp32 path reproduces the golden to 1e-5 relative)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-4)
    grads = dict(m.named_parameters())
    for name, ref in zip(g["grad_names"].tolist(), g["grad_norms"].tolist()):
        assert grads[name].grad.double().norm().item() == pytest.approx(ref, rel=3e-2), name


@pytest.mark.parametrize("tag,beta,fn_", [("a", 1.0, 2.0), ("b", 0.3, 0.0)])
def test_vrnn_small_bf16_against_reference_tensors(tag, beta, fn_):
    """Every tensor the small golden holds (z, h_n, all gradients) at bf16-operand distance from the reference's fp32 values
    (widths 8 / 32 / 16: few terms per product, so the rounding of single operands shows — gradients of ~1e-6 magnitude move by
    up to ~10 %; at the C2 widths the per-parameter gradient norms stay within 3 %, see the test above)."""
    g = np.load(os.path.join(GOLDEN, "vrnn_small.npz"))
    m = VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, num_mix=10, num_bins=2**16)
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd.")})
    m.to(DEV)
    x, x_sl, eps = T(g["x"]), T(g["x_sl"]), T(g[f"{tag}_eps"])
    loss, metrics, out = _run(m, x, x_sl, eps, beta, fn_)
    assert float(loss) == pytest.approx(float(g[f"{tag}_loss"]), rel=2e-3)
    assert rel_l2(out.z, T(g[f"{tag}_z"])) < 2e-2
    assert rel_l2(out.h_n, T(g[f"{tag}_h_n"])) < 2e-2
    assert rel_l2(out.z, T(g[f"{tag}_z"])) > 1e-6  # not the fp32 path
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, T(g[f"{tag}_grad.{k}"])) < 0.15, k


def test_bf16_and_f32_steps_agree_on_headline_shape():
    """[16, 16000] at the headline widths: one train-step's loss in both modes on the same weights, noise and batch."""
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(DEV)
    x, x_sl = O.synth_batch(16, 16000, seed=1, ragged=True)
    eps = torch.randn(250, 16, 256, generator=torch.Generator().manual_seed(5))
    lb, _, ob = _run(m, x, x_sl, eps, 1.0, 0.0)
    gb = [p.grad.clone() for p in m.parameters()]
    _hip.set_operand_dtype("f32")
    lf, _, of = _run(m, x, x_sl, eps, 1.0, 0.0)
    gf = [p.grad.clone() for p in m.parameters()]
    _hip.set_operand_dtype("bf16")
    frames = x_sl.double()
    d = ((ob.elbo.cpu().double() - of.elbo.cpu().double()).abs() / frames).max()
    assert 0.0 < float(d) < NATS_PER_FRAME_BUDGET, float(d)
    cos = sum(float((a.double() * b.double()).sum()) for a, b in zip(gb, gf)) / (
        sum(float(a.double().pow(2).sum()) for a in gb) ** 0.5 * sum(float(b.double().pow(2).sum()) for b in gf) ** 0.5)
    assert cos > 0.995, cos


def test_srnn_bf16_delta_within_budget():
    torch.manual_seed(0)
    m = SRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, smoothing=True).to(DEV)
    x, x_sl = O.synth_batch(8, 6400, seed=2, ragged=True)
    Tp = 100
    eps = torch.randn(Tp, 8, 256, generator=torch.Generator().manual_seed(7))

    def run():
        m.zero_grad()
        loss, _, out = m(x.to(DEV), x_sl, beta=1.0, free_nats=0.0, eps=eps.to(DEV))
        loss.backward()
        return float(loss), out.elbo.cpu().double()

    lb, eb = run()
    _hip.set_operand_dtype("f32")
    lf, ef = run()
    _hip.set_operand_dtype("bf16")
    d = ((eb - ef).abs() / x_sl.double()).max()
    assert 0.0 < float(d) < NATS_PER_FRAME_BUDGET, float(d)


def test_wavenet_c5_dims_bf16_delta_within_budget():
    """BASELINE configs[4] (the reference runs WaveNet under fp16 autocast): C5 dims on the golden inputs of
    `test_wavenet_c5_dims_vs_reference_golden` with the block kernels' products on the bf16 matrix pipe (fused forward / backward /
    weight-gradient kernels of K10 and the K6 GEMMs around them), against the REFERENCE's fp32 values."""
    from blvm.models import WaveNet
    from blvm.modules.distributions import DiscretizedLogisticMixtureDense

    g = np.load(os.path.join(GOLDEN, "wavenet.npz"))
    torch.manual_seed(0)
    lik = DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16)
    m = WaveNet(likelihood=lik, n_layers=10, n_stacks=5, res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1).to(DEV)
    x, x_sl = O.synth_batch(2, 1500, seed=0, ragged=True)
    loss, metrics, out = m(x.to(DEV), x_sl)
    loss.backward()
    d = ((out.log_prob.detach().cpu().double() - T(g["f_log_prob"]).double()).abs() / x_sl.double()).max()
    assert 0.0 < float(d) < NATS_PER_FRAME_BUDGET, float(d)
    assert float(loss) == pytest.approx(float(g["f_loss"]), rel=1e-4)
    grads = dict(m.named_parameters())
    for name, ref in zip(g["f_grad_names"].tolist(), g["f_grad_norms"].tolist()):
        assert grads[name].grad.double().norm().item() == pytest.approx(ref, rel=5e-2), name


def test_one_launch_decoders_in_bf16_mode():
    """The sampling programs with bf16 weight packs: finite, in range, and — where no mixture pick flips — close to the fp32 roll-out."""
    B, T_ = 8, 6
    g = torch.Generator().manual_seed(13)
    torch.manual_seed(23)
    v = VRNNAudio(likelihood="DMoL", input_size=16, hidden_size=32, latent_size=16, residual_posterior=True).to(DEV)
    eps = torch.randn(T_, B, 16, generator=g).to(DEV)
    uni = (torch.empty(T_, B, 16, 10).uniform_(1e-5, 1 - 1e-5, generator=g).to(DEV), torch.empty(T_, B, 16).uniform_(1e-8, 1 - 1e-8, generator=g).to(DEV))
    x0 = (torch.rand(B, 16, 1, generator=g) * 0.2 - 0.1).to(DEV)
    (b, _), _ = v.generate(n_samples=B, max_timesteps=T_, x=x0, eps=eps, uniforms=uni, fused=True)
    _hip.set_operand_dtype("f32")
    (a, _), _ = v.generate(n_samples=B, max_timesteps=T_, x=x0, eps=eps, uniforms=uni, fused=True)
    _hip.set_operand_dtype("bf16")
    assert torch.isfinite(b).all() and float(b.abs().max()) <= 1.0
    assert float(((a[:, :2] - b[:, :2]).abs() > 2e-2).float().mean()) < 0.1  # the first generated stack: one step of bf16 products
    assert not torch.equal(a, b)
    _hip.check_async()
