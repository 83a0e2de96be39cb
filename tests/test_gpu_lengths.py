"""GPU (MI355X): the sequence lengths BASELINE.json's configs are quoted on — VRNN `[32, 49152]` (T' = 768,
"TIMIT-length") and SRNN `[16, 196608]` (T' = 3072, "LibriSpeech-length") — which no golden covers: reserve / workspace sizing and
64-bit offsets at 3-12x the golden-covered sequence length, through size-independent properties (row independence, invariance to
samples beyond x_sl, the bits/dim window), agreement of the two execution paths of the recurrent chains (one persistent launch vs
one launch per link), one oracle comparison at T' = 768 on a reduced width, and the struct-argument fallback of the packed
link kernels (csrc/stages.h) that only shapes outside the packed fields' ranges reach.
Reference lines: blvm/models/vrnn.py:281-369, blvm/models/srnn.py:162-302."""
import pytest
import torch

import blvm_oracle as O
from blvm import _hip
from blvm.models import SRNNAudio, VRNNAudio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_hip():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert _hip.load().blvm_device_ok() == 1, "libblvm_hip: no gfx950 device visible"
    yield
    _hip.load().blvm_pchain_configure(128, -1)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _train_step(m, x, x_sl, eps, persistent):
    _hip.load().blvm_pchain_configure(128 if persistent else 0, -1)
    for p in m.parameters():
        p.grad = None
    loss, metrics, out = m(x, x_sl, beta=1.0, free_nats=2.0, eps=eps)
    loss.backward()
    torch.cuda.synchronize()
    _hip.check_async()
    return loss.detach(), metrics, out, {k: p.grad.clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize("cls,B,T_", [(VRNNAudio, 32, 49152), (SRNNAudio, 16, 196608)])
def test_baseline_sequence_lengths_properties_and_both_execution_paths(cls, B, T_):
    torch.manual_seed(0)
    kw = dict(smoothing=True) if cls is SRNNAudio else {}
    m = cls(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, **kw).to(DEV)
    x, x_sl = O.synth_batch(B, T_, seed=0, ragged=True)
    Tp = T_ // 64
    eps = torch.randn(Tp, B, 256, generator=torch.Generator().manual_seed(1)).to(DEV)
    xd = x.to(DEV)
    loss_p, metrics, full, g_p = _train_step(m, xd, x_sl, eps, persistent=True)
    loss_l, _, full_l, g_l = _train_step(m, xd, x_sl, eps, persistent=False)
    # (0) one persistent launch per sequence == one launch per link, to fp32 summation order
    assert float(loss_p) == pytest.approx(float(loss_l), rel=1e-6)
    torch.testing.assert_close(full.elbo, full_l.elbo, rtol=1e-6, atol=0)
    for k in g_p:
        assert torch.isfinite(g_p[k]).all(), k
        assert rel_l2(g_p[k], g_l[k]) < 1e-3, k
    # (i) a sub-batch gives the same per-utterance terms; (ii) samples beyond x_sl influence nothing
    sub = slice(B // 4, B // 4 + B // 2)
    Ts = int(x_sl[sub].max())
    Tps = (Ts + 63) // 64
    xs = x[sub, :Ts].clone()
    for i, n in enumerate(x_sl[sub].tolist()):
        xs[i, ((n + 63) // 64) * 64 :] = 0.77
    with torch.no_grad():
        _, _, part = m(xs.to(DEV), x_sl[sub], beta=1.0, free_nats=2.0, eps=eps[:Tps, sub].contiguous())
    torch.testing.assert_close(part.elbo, full.elbo[sub], rtol=2e-6, atol=0)
    torch.testing.assert_close(part.kl, full.kl[sub], rtol=2e-6, atol=0)
    # (iii) bits/dim at random init
    bpd = {mm.name: mm.value for mm in metrics}["bpd"]
    assert 16.5 < bpd < 18.0


def test_vrnn_768_steps_vs_oracle_reduced_width():
    """T' = 768 recurrent steps against the CPU oracle (narrow model so that the oracle finishes in seconds): the recurrence does
    not drift over a TIMIT-length sequence.  Ragged, B not a multiple of 16, both execution paths."""
    torch.manual_seed(3)
    S, Hd, Z, B, Tp = 8, 32, 16, 3, 768
    m = VRNNAudio(likelihood="DMoL", input_size=S, hidden_size=Hd, latent_size=Z, residual_posterior=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, x_sl = O.synth_batch(B, S * Tp - 5, seed=4, ragged=True)
    eps = torch.randn(Tp, B, Z, generator=torch.Generator().manual_seed(5))
    ref = O.vrnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=2.0, stack=S)
    ref["loss"].backward()
    m.to(DEV)
    for persistent in (True, False):
        loss, _, out, g = _train_step(m, x.to(DEV), x_sl, eps.to(DEV), persistent)
        assert float(loss) == pytest.approx(float(ref["loss"]), rel=1e-5)
        torch.testing.assert_close(out.elbo.cpu(), ref["elbo"].detach(), rtol=1e-5, atol=1e-3)
        for k in g:
            assert rel_l2(g[k], sd[k].grad) < 1e-3, (persistent, k)


@pytest.mark.parametrize("model", ["vrnn", "srnn"])
def test_struct_argument_fallback_of_the_packed_link_kernels_vs_oracle(model):
    """The scalar-argument link kernels pack B, lda, K and tile counts into 12/16-bit fields and fall back to the struct-argument
    kernels when a value does not fit (csrc/stages.h launch_lin_n).  B = 4100 rows with 48-wide layers (not a multiple of 32, so the
    32x32 large-batch kernel does not apply either) is out of the 12-bit batch field: the three-segment first link runs on
    lin_stage_kernel<.., 3>, the symmetric two-segment links on linp2_stage_kernel instead of lin2s.  Same numbers as the oracle."""
    torch.manual_seed(8)
    B, S, Tp, Hd, Z = 4100, 16, 3, 48, 16
    T_ = S * Tp - 2
    cls = VRNNAudio if model == "vrnn" else SRNNAudio
    m = cls(likelihood="DMoL", input_size=S, hidden_size=Hd, latent_size=Z, residual_posterior=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, x_sl = O.synth_batch(B, T_, seed=13, ragged=True)
    eps = torch.randn(Tp, B, Z, generator=torch.Generator().manual_seed(2))
    fwd = O.vrnn_audio_forward if model == "vrnn" else O.srnn_audio_forward
    ref = fwd(sd, x, x_sl, eps, beta=0.8, free_nats=1.0, stack=S)
    ref["loss"].backward()
    m.to(DEV)
    loss, _, out = m(x.to(DEV), x_sl, beta=0.8, free_nats=1.0, eps=eps.to(DEV))
    loss.backward()
    assert float(loss) == pytest.approx(float(ref["loss"]), rel=1e-5)
    torch.testing.assert_close(out.elbo.cpu(), ref["elbo"].detach(), rtol=1e-5, atol=1e-3)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, sd[k].grad) < 1e-3, k


@pytest.mark.parametrize("B,T_", [(256, 64 * 20), (200, 64 * 12 - 7), (128, 64 * 30), (72, 64 * 9 + 5)])
def test_vrnn_row_group_engine_equals_launch_per_link(B, T_):
    """65 <= B <= 256 (round 3): the persistent programs on ROW GROUPS of two row tiles (csrc/pchain_rt.h) against the launch-per-link
    path on the same inputs — headline widths, ragged lengths, batches whose last group is partial (200 rows = 13 row tiles = 6
    groups + 1 tile; 72 rows = 2 groups + half a row tile) — loss, per-utterance ELBO / KL, latents, the final state and every
    gradient, to fp32 summation order."""
    torch.manual_seed(0)
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(DEV)
    x, x_sl = O.synth_batch(B, T_, seed=2, ragged=True)
    Tp = (T_ + 63) // 64
    eps = torch.randn(Tp, B, 256, generator=torch.Generator().manual_seed(1)).to(DEV)
    loss_p, _, out_p, g_p = _train_step(m, x.to(DEV), x_sl, eps, persistent=True)
    loss_l, _, out_l, g_l = _train_step(m, x.to(DEV), x_sl, eps, persistent=False)
    assert float(loss_p) == pytest.approx(float(loss_l), rel=1e-6)
    torch.testing.assert_close(out_p.elbo, out_l.elbo, rtol=1e-6, atol=0)
    torch.testing.assert_close(out_p.kl, out_l.kl, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out_p.z, out_l.z, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out_p.h_n, out_l.h_n, rtol=1e-4, atol=1e-5)
    for k in g_p:
        assert torch.isfinite(g_p[k]).all(), k
        assert rel_l2(g_p[k], g_l[k]) < 1e-3, k


def test_vrnn_row_group_engine_vs_oracle():
    """The row-group engine against the CPU oracle on a narrow model: B = 150 (10 row tiles, the last one partial), ragged, T not a
    multiple of the stack."""
    torch.manual_seed(4)
    S, Hd, Z, B, Tp = 8, 32, 16, 150, 9
    m = VRNNAudio(likelihood="DMoL", input_size=S, hidden_size=Hd, latent_size=Z, residual_posterior=True)
    sd = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, x_sl = O.synth_batch(B, S * Tp - 3, seed=6, ragged=True)
    eps = torch.randn(Tp, B, Z, generator=torch.Generator().manual_seed(7))
    ref = O.vrnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=2.0, stack=S)
    ref["loss"].backward()
    m.to(DEV)
    loss, _, out, g = _train_step(m, x.to(DEV), x_sl, eps.to(DEV), persistent=True)
    assert float(loss) == pytest.approx(float(ref["loss"]), rel=1e-5)
    torch.testing.assert_close(out.elbo.cpu(), ref["elbo"].detach(), rtol=1e-5, atol=1e-3)
    for k in g:
        assert rel_l2(g[k], sd[k].grad) < 1e-3, k
