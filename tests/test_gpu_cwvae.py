"""GPU (MI355X): Clockwork-VAE (BASELINE config C4) through the C ABI against
  (1) golden vectors produced by the imported reference (tests/golden/cwvae.npz, oracle/gen_golden.py::gen_cwvae),
  (2) the CPU oracle (oracle/blvm_oracle.py::cwvae_audio_forward) evaluated in float64, and
  (3) size-independent properties at BASELINE's C4 size.
Tolerances: loss / ELBO / log-likelihood 1e-4 relative (north_star; observed ~1e-6); latents 1e-4 absolute; gradients by
relative L2 against the float64 oracle: no further than max(2 x the reference's own fp32 distance, 1e-3)."""
import os

import numpy as np
import pytest
import torch

import blvm_oracle as O
from blvm import _hip
from blvm.models import CWVAEAudio

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CW_SMALL = dict(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
                num_mix=10, num_bins=2**16)  # fmt: skip
CW_FULL = dict(z_size=[128, 64, 32], h_size=192, strides=[64, 16, 16], num_level_layers=8, stride_per_layer=2,
               precision_posterior=True, likelihood="DMoL", num_bins=2**16)  # fmt: skip


@pytest.fixture(scope="module", autouse=True)
def _require_hip():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert _hip.load().blvm_device_ok() == 1


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "cwvae.npz"))


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def close(a, b, rtol, atol=0.0):
    torch.testing.assert_close(a.detach().double().cpu(), (b if isinstance(b, torch.Tensor) else T(b)).double(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,kw,beta,fn_", [("pw", dict(precision_posterior=True), 1.0, 0.5), ("rs", dict(residual_posterior=True), 0.7, 0.0)])
def test_cwvae_small_matches_reference(g, tag, kw, beta, fn_):
    m = CWVAEAudio(**CW_SMALL, **kw)
    pre = f"{tag}_sd."
    sd = {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m = m.to(DEV)
    x, x_sl = T(g["x"]), T(g["x_sl"])
    eps = [T(g[f"{tag}_eps{l}"]).to(DEV) for l in range(3)]
    loss, metrics, o = m(x.to(DEV), x_sl, beta=beta, free_nats=fn_, eps=eps)
    loss.backward()

    close(loss, g[f"{tag}_loss"], 1e-4)
    close(o.elbo, g[f"{tag}_elbo"], 1e-4)
    close(o.log_prob, g[f"{tag}_log_prob"], 1e-4)
    close(o.kld, g[f"{tag}_kld"], 1e-4, 1e-5)
    for l in range(3):
        close(o.z[l], g[f"{tag}_z{l}"], 1e-4, 1e-4)
        close(o.enc_mus[l], g[f"{tag}_enc_mu{l}"], 1e-4, 1e-4)
        close(o.prior_mus[l], g[f"{tag}_prior_mu{l}"], 1e-4, 1e-4)
        close(o.state_n[l][0], g[f"{tag}_state_z{l}"], 1e-4, 1e-4)
        close(o.state_n[l][1], g[f"{tag}_state_h{l}"], 1e-4, 1e-4)
        assert torch.equal(o.z_sl[l].to(torch.int64), T(g[f"{tag}_z_sl{l}"]).to(torch.int64))
    close(o.reconstructions_parameters[0], g[f"{tag}_params"], 1e-3, 1e-4)
    assert [mm.name for mm in metrics] == list(g[f"{tag}_metric_names"])
    np.testing.assert_allclose([mm.value for mm in metrics], g[f"{tag}_metric_values"], rtol=1e-4, atol=1e-6)

    # gradients against the float64 oracle, with the reference's own fp32 gradients as the yardstick
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    out64 = O.cwvae_audio_forward(sd64, x.double(), x_sl, [e.double().cpu() for e in eps], beta=beta, free_nats=fn_,
                                  strides=CW_SMALL["strides"], num_level_layers=2, stride_per_layer=2, num_bins=2**16, **kw)
    out64["loss"].backward()
    for k, p in m.named_parameters():
        truth = sd64[k].grad
        assert p.grad is not None, k
        ref_err = rel(T(g[f"{tag}_grad.{k}"]), truth)
        assert rel(p.grad, truth) <= max(2 * ref_err, 1e-3), (k, rel(p.grad, truth), ref_err)


def test_cwvae_carried_state(g):
    """Second call that starts every level from the previous call's state_n (experiment_clockwork_audio.py:263-271)."""
    m = CWVAEAudio(**CW_SMALL, precision_posterior=True)
    m.load_state_dict({k[6:]: T(g[k]) for k in g.files if k.startswith("pw_sd.")})
    m = m.to(DEV)
    x, x_sl = T(g["x"]).to(DEV), T(g["x_sl"])
    _, _, o = m(x, x_sl, beta=1.0, free_nats=0.5, eps=[T(g[f"pw_eps{l}"]).to(DEV) for l in range(3)])
    state0 = [(z.detach().contiguous(), h.detach().contiguous()) for z, h in o.state_n]
    loss2, _, o2 = m(x, x_sl, state0=state0, beta=1.0, free_nats=0.5, eps=[T(g[f"c_eps{l}"]).to(DEV) for l in range(3)])
    close(loss2, g["c_loss"], 1e-4)
    close(o2.elbo, g["c_elbo"], 1e-4)
    for l in range(3):
        close(o2.z[l], g[f"c_z{l}"], 1e-4, 1e-4)
    loss2.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_cwvae_with_resets_matches_reference():
    """CWVAE(with_resets=True) (`clockwork_vae.py:273-275`: every level below the top restarts from a zero state whenever its
    parent ticks) against the reference's own outputs (tests/golden/cwvae_resets.npz): the HIP path runs the segments between two
    resets as rows of one short sequence launch.  Level 1 has 19 steps for a parent stride of 2 (a padded last segment); ragged
    lengths, free nats 0.5; loss / ELBO / KL, latents, the carried states and every parameter gradient; then generation."""
    g = np.load(os.path.join(GOLDEN, "cwvae_resets.npz"))
    m = CWVAEAudio(**CW_SMALL, precision_posterior=True)
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd.")})
    m.cwvae.with_resets = True
    m = m.to(DEV)
    eps = [T(g[f"eps{l}"]).to(DEV) for l in range(3)]
    loss, _, o = m(T(g["x"]).to(DEV), T(g["x_sl"]), beta=1.0, free_nats=0.5, eps=eps)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-5)
    close(o.elbo, g["elbo"], 1e-5, 1e-3)
    close(o.log_prob, g["log_prob"], 1e-5, 1e-3)
    close(o.kld, g["kld"], 1e-4, 1e-3)
    for l in range(3):
        close(o.z[l], g[f"z{l}"], 1e-4, 1e-5)
        close(o.state_n[l][0], g[f"state_z{l}"], 1e-4, 1e-5)
        close(o.state_n[l][1], g[f"state_h{l}"], 1e-4, 1e-5)
    # gradients against the float64 oracle, with the reference's own fp32 gradients as the yardstick (as in the test above)
    sd64 = {k[3:]: T(g[k]).double().requires_grad_(True) for k in g.files if k.startswith("sd.")}
    out64 = O.cwvae_audio_forward(sd64, T(g["x"]).double(), T(g["x_sl"]), [e.double().cpu() for e in eps], beta=1.0, free_nats=0.5,
                                  strides=CW_SMALL["strides"], num_level_layers=2, stride_per_layer=2, num_bins=2**16,
                                  precision_posterior=True, with_resets=True)  # fmt: skip
    out64["loss"].backward()
    for k, p in m.named_parameters():
        truth = sd64[k].grad
        assert p.grad is not None, k
        ref_err = rel(T(g[f"grad.{k}"]), truth)
        assert rel(p.grad, truth) <= max(2 * ref_err, 1e-3), (k, rel(p.grad, truth), ref_err)
    geps = [T(g[f"gen_eps{l}"]).to(DEV) for l in range(3)]
    (x, _), _ = m.generate(n_samples=2, max_timesteps=int(g["gen_T"][0]), use_mode_observations=True, eps=geps)
    assert tuple(x.shape) == tuple(g["gen_x"].shape)
    close(x, g["gen_x"], 1e-4, 1e-5)


def _full_model():
    torch.manual_seed(0)
    return CWVAEAudio(**CW_FULL)


def test_cwvae_c4_dims_match_reference(g):
    """BASELINE config C4 dimensions; weights reproduced from the seed (same construction order), pinned by checksums."""
    m = _full_model()
    assert list(m.state_dict().keys()) == list(g["param_names"])
    for k, v in m.state_dict().items():
        c = g[f"cks.{k}"]
        assert tuple(v.shape) == tuple(int(s) for s in c[2:]), k
        np.testing.assert_allclose([v.double().sum().item(), v.double().abs().sum().item()], c[:2], rtol=1e-9, atol=1e-9, err_msg=k)
    m = m.to(DEV)
    x, _ = O.synth_batch(2, 16384, seed=0)
    x_sl = T(g["f_x_sl"])
    T_l = [int(v) for v in g["f_T_l"]]
    torch.manual_seed(123)  # the reference's draw order: top level first, one randn(B, z) per step
    eps = [None] * 3
    for l in (2, 1, 0):
        eps[l] = torch.stack([torch.randn(2, CW_FULL["z_size"][l]) for _ in range(T_l[l])], 0).to(DEV)
    loss, metrics, o = m(x.to(DEV), x_sl, beta=1.0, free_nats=4.0, eps=eps)
    loss.backward()
    # At these dimensions the fp32 network is ill-conditioned (per-channel norms over time of nearly constant up-sampled
    # contexts cancel catastrophically): the reference's OWN fp32 results sit 1.6e-4 (loss) and 1e-2..7e-2 (latents) away
    # from a float64 evaluation of the same graph.  So everything is measured against the float64 oracle: the HIP path may
    # not be further from it than twice the reference (and is within the 1e-4 north-star tolerance of the reference or of
    # float64, whichever is nearer).
    sd64 = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        o64 = O.cwvae_audio_forward(sd64, x.double(), x_sl, [e.double().cpu() for e in eps], beta=1.0, free_nats=4.0,
                                    strides=CW_FULL["strides"], num_level_layers=8, stride_per_layer=2, num_bins=2**16,
                                    precision_posterior=True)  # fmt: skip
    for name, hip, ref, truth in (("loss", loss, g["f_loss"], o64["loss"]), ("elbo", o.elbo, g["f_elbo"], o64["elbo"]),
                                  ("log_prob", o.log_prob, g["f_log_prob"], o64["log_prob"]), ("kld", o.kld, g["f_kld"], o64["kld"])):  # fmt: skip
        hip, ref, truth = hip.detach().double().cpu().reshape(-1), T(ref).double().reshape(-1), truth.double().reshape(-1)
        hip_err, ref_err = float(((hip - truth) / truth).abs().max()), float(((ref - truth) / truth).abs().max())
        assert hip_err <= max(2 * ref_err, 1e-4), (name, hip_err, ref_err)
        assert min(hip_err, float(((hip - ref) / ref).abs().max())) <= (2e-3 if name == "kld" else 2e-4), (name, hip_err)
    for l in range(3):
        truth = o64["z"][l].transpose(0, 1)
        ref_err = float((T(g[f"f_z{l}"]).double() - truth).abs().max())
        hip_err = float((o.z[l].detach().double().cpu() - truth).abs().max())
        assert hip_err <= max(2 * ref_err, 1e-4), (l, hip_err, ref_err)
    assert [mm.name for mm in metrics] == list(g["f_metric_names"])
    for mm, ref_v in zip(metrics, g["f_metric_values"]):  # per-level KLs inherit the latents' fp32 noise (see above)
        np.testing.assert_allclose(mm.value, ref_v, rtol=2e-3 if mm.name.startswith("kl") else 3e-4, atol=1e-6, err_msg=mm.name)
    # Gradients at these dimensions are not compared with the reference: at random init the C4 network is chaotic in fp32
    # (the reference's own fp32 decoder activations differ ~50 % from a float64 evaluation of the same graph, its gradients
    # by > 100 %; measured with oracle fp32 vs fp64, DESIGN.md "CW-VAE conditioning"), and with a single top-level step every
    # norm of the top decoder is degenerate (variance exactly 0).  Gradient parity is pinned on the reduced model above and,
    # at C4 widths, per block / per cell below and in test_gpu_convcoder.py.
    assert [k for k, _ in m.named_parameters()] == list(g["f_grad_names"])
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_rssm_cell_c4_dims_vs_oracle_float64():
    """One CW-VAE level at C4 widths (h=192, z=64, context 192, 48 steps, ragged lengths, free nats, precision-weighted
    posterior) against the oracle cell stepped in float64: states, KL and every gradient."""
    from blvm.modules.rssm import RSSMCell

    torch.manual_seed(5)
    Tn, B, H, Z, C, E, stride, fn = 48, 8, 192, 64, 192, 192, 1024, 4.0 * 16
    cell = RSSMCell(z_dim=Z, h_dim=H, c_dim=C, e_dim=E, precision_posterior=True)
    gen = torch.Generator().manual_seed(6)
    enc, ctx = torch.randn(Tn, B, E, generator=gen), torch.randn(Tn, B, C, generator=gen)
    eps = torch.randn(Tn, B, Z, generator=gen)
    z0, h0 = 0.3 * torch.randn(B, Z, generator=gen), 0.3 * torch.randn(B, H, generator=gen)
    wz, wh = torch.randn(Tn, B, Z, generator=gen), torch.randn(Tn, B, H, generator=gen)
    x_sl = torch.tensor([49152, 49152, 40000, 33333, 30000, 20000, 16385, 1000])

    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in cell.state_dict().items()}
    leaves = [t.double().requires_grad_(True) for t in (enc, ctx, z0, h0)]
    zs, hs, d = O.rssm_sequence(sd64, leaves[0], leaves[1], (leaves[2], leaves[3]), eps.double(), precision_posterior=True)
    kl = O.kl_gaussian(d["enc_mu"], d["enc_sd"], d["prior_mu"], d["prior_sd"])
    mask = O.sequence_mask(torch.ceil(x_sl / stride).long(), max_len=Tn).t().unsqueeze(-1)
    kld_fn = (O.discount_free_nats(kl, fn) * mask).sum((0, 2))
    ((zs * wz.double()).sum() + (hs * wh.double()).sum() + kld_fn.sum()).backward()

    cell = cell.to(DEV)
    dl = [t.to(DEV).requires_grad_(True) for t in (enc, ctx, z0, h0)]
    zs_d, hs_d, kld_d, kld_fn_d, *_ = cell.sequence(dl[0], dl[1], (dl[2], dl[3]), eps.to(DEV), x_sl.to(DEV, dtype=torch.int32), stride, fn)
    ((zs_d[1:] * wz.to(DEV)).sum() + (hs_d[1:] * wh.to(DEV)).sum() + kld_fn_d.sum()).backward()
    assert rel(zs_d[1:], zs) < 1e-5 and rel(hs_d[1:], hs) < 1e-5
    assert rel(kld_fn_d, kld_fn) < 1e-5 and rel(kld_d, (kl * mask).sum((0, 2))) < 1e-5
    for name, a, b in zip(("enc", "ctx", "z0", "h0"), dl, leaves):
        assert rel(a.grad, b.grad) < 1e-4, (name, rel(a.grad, b.grad))
    for k, p in cell.named_parameters():
        assert rel(p.grad, sd64[k].grad) < 1e-4, (k, rel(p.grad, sd64[k].grad))


def test_cwvae_c4_full_size_rows_are_independent():
    """BASELINE C4 per-GPU shape [8, 49152]: every op is per-utterance (the norm is per sample and channel), so the ELBO of a
    row must not depend on what else is in the batch — this is what makes the batch dimension shardable."""
    m = _full_model().to(DEV)
    B, Tn = 8, 49152
    x, _ = O.synth_batch(B, Tn, seed=3)
    x_sl = torch.tensor([49152, 49152, 40000, 33333, 30000, 20000, 16385, 1000])
    x = (x * (torch.arange(Tn).unsqueeze(0) < x_sl.unsqueeze(1))).to(DEV)
    T_l = [768, 48, 3]
    gen = torch.Generator(device=DEV).manual_seed(9)
    eps = [torch.randn(T_l[l], B, CW_FULL["z_size"][l], device=DEV, generator=gen) for l in range(3)]
    loss, metrics, o = m(x, x_sl, beta=1.0, free_nats=4.0, eps=eps)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters())
    bpd = {mm.name: mm.value for mm in metrics}["elbo (bpt)"]
    assert 16.5 < bpd < 18.0, bpd  # random init on 16-bit mu-law: ~log2(65536) + 1
    assert [tuple(z.shape) for z in o.z] == [(B, 768, 128), (B, 48, 64), (B, 3, 32)]
    assert [s.tolist() for s in o.z_sl] == [[768, 768, 625, 521, 469, 313, 257, 16], [48, 48, 40, 33, 30, 20, 17, 1], [3, 3, 3, 3, 2, 2, 2, 1]]
    rows = [2, 7]
    with torch.no_grad():
        _, _, o2 = m(x[rows].contiguous(), x_sl[rows], beta=1.0, free_nats=4.0, eps=[e[:, rows].contiguous() for e in eps])
    close(o2.elbo, o.elbo[rows].cpu(), 1e-5)
    close(o2.kld, o.kld[rows].cpu(), 1e-4, 1e-4)


def test_cwvae_generate_matches_reference():
    """CWVAEAudio.generate (ancestral sampling, mode of the observation model) against the reference's own output for the
    same prior draws (tests/golden/generate.npz); output length 205 for max_timesteps=64 is the reference's."""
    g = np.load(os.path.join(GOLDEN, "generate.npz"))
    m = CWVAEAudio(**CW_SMALL, precision_posterior=True)
    m.load_state_dict({k[6:]: T(g[k]) for k in g.files if k.startswith("cw_sd.")})
    m = m.to(DEV)
    eps = [T(g[f"cw_eps{l}"]).to(DEV) for l in range(3)]
    (x, x_sl), _ = m.generate(n_samples=2, max_timesteps=int(g["cw_T"][0]), use_mode_observations=True, eps=eps)
    assert tuple(x.shape) == tuple(g["cw_x_mode"].shape)
    close(x, g["cw_x_mode"], 1e-4, 1e-5)
    assert torch.equal(x_sl, T(g["cw_x_sl"]).to(x_sl.dtype))
    (xs, _), _ = m.generate(n_samples=3, max_timesteps=64)  # stochastic observations, device RNG
    assert tuple(xs.shape) == (3, 205, 1) and torch.isfinite(xs).all() and float(xs.abs().max()) <= 1.0
