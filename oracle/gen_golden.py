"""Generate golden vectors by importing the UNMODIFIED reference (build container only).

    BLVM_DATA_ROOT_DIRECTORY=/tmp/blvm_data PYTHONDONTWRITEBYTECODE=1 \
    PYTHONPATH=oracle/refshim:/root/reference python oracle/gen_golden.py

Writes small `.npz` fixtures (inputs + expected outputs only — no reference source) into `tests/golden/`.
The reference never travels to the GPU box; the fixtures do.  `oracle/refshim/` holds annotation/logging
no-op modules for third-party packages the reference imports but that are off the arithmetic path.

TEST INFRASTRUCTURE: not imported by the product.
"""
import math
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)

import blvm  # noqa: E402  (the reference, from /root/reference via PYTHONPATH)
import blvm.models as RM  # noqa: E402
from blvm.data.transforms import MuLawEncode  # noqa: E402
from blvm.modules.distributions import DiagonalGaussianDense, DiscretizedLogisticMixtureDense  # noqa: E402
from blvm.training.annealers import CosineAnnealer  # noqa: E402
from blvm.utils import log_likelihoods as RLL  # noqa: E402
from blvm.utils import operations as ROP  # noqa: E402
from blvm.utils import variational as RV  # noqa: E402

import blvm_oracle as O  # noqa: E402  (only for the input synthesiser, so inputs are defined in ONE place)

assert "/root/reference" in blvm.__file__, blvm.__file__


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def gen_functions():
    g = torch.Generator().manual_seed(7)
    R = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    out = {}

    # DMoL log-likelihood incl. edge cases (y = +-1, near edges, tiny scales -> delta fallback, clamp at -7)
    N, K = 64, 10
    y = torch.rand(N, 1, generator=g) * 1.9 - 0.95
    y[0], y[1], y[2], y[3] = 1.0, -1.0, 1 - 1.0 / 65536, -1 + 1.0 / 65536
    logits, locs, ls = R(N, K), R(N, 1, K) * 0.5, R(N, 1, K) * 2 - 4
    ls[4:8] = -7.0  # clamp floor: sharp components -> delta < 1e-5 fallback when far from loc
    ls[8:10] = 3.0
    out.update(dmol_y=y, dmol_logits=logits, dmol_locs=locs, dmol_ls=ls)
    out["dmol_ll_65536"] = RLL.discretized_logistic_mixture_ll(y, logits, locs, ls, num_bins=2**16)
    out["dmol_ll_256"] = RLL.discretized_logistic_mixture_ll(y, logits, locs, ls, num_bins=256)

    # DMoL head (Linear 30->30, split, clamp)
    torch.manual_seed(3)
    head = DiscretizedLogisticMixtureDense(x_dim=30, y_dim=1, num_mix=10, num_bins=2**16)
    xh = R(5, 7, 30) * 3
    lg, lc, lsc = head(xh)
    out.update(dmolhead_w=head.params.weight, dmolhead_b=head.params.bias, dmolhead_x=xh, dmolhead_logits=lg, dmolhead_locs=lc, dmolhead_ls=lsc)
    out["dmolhead_mode"] = head.mode((lg, lc, lsc))

    # Gaussian head (softplus beta) incl. values beyond the softplus threshold
    torch.manual_seed(4)
    gh = DiagonalGaussianDense(12, 6)
    xg = R(9, 12) * 30
    mu, sd = gh(xg)
    out.update(ghead_w=gh.params.weight, ghead_b=gh.params.bias, ghead_x=xg, ghead_mu=mu, ghead_sd=sd)

    # Gaussian / GMM ll
    yg, mug, sdg = R(6, 5), R(6, 5), R(6, 5).abs() + 1e-3
    out.update(gll_y=yg, gll_mu=mug, gll_sd=sdg, gll_eps0=RLL.gaussian_ll(yg, mug, sdg, epsilon=0, reduce_dim=None))
    ym, lgm, mum, sdm = R(6, 1), R(6, 4), R(6, 1, 4), R(6, 1, 4).abs() + 1e-3
    out.update(gmm_y=ym, gmm_logits=lgm, gmm_mu=mum, gmm_sd=sdm, gmm_ll=RLL.gaussian_mixture_ll(ym, lgm, mum, sdm, epsilon=1e-4))

    # KL, free nats, precision weighting
    mq, sq, mp, sp = R(4, 6, 8), R(4, 6, 8).abs() + 0.05, R(4, 6, 8), R(4, 6, 8).abs() + 0.05
    kl = RV.kl_divergence_gaussian(mq, sq, mp, sp)
    out.update(kl_mq=mq, kl_sq=sq, kl_mp=mp, kl_sp=sp, kl_out=kl)
    out["kl_fn2"] = RV.discount_free_nats(kl, 2.0, shared_dims=-1)
    out["kl_fn0"] = RV.discount_free_nats(kl, 0, shared_dims=-1)
    pm, ps = RV.precision_weighted_gaussian(mq, sq, mp, sp)
    out.update(pw_mu=pm, pw_sd=ps)

    # stack / mask / reverse
    xs = R(3, 21)
    st, pad = ROP.stack_tensor(xs, 8, dim=1)
    out.update(stack_x=xs, stack_out=st, stack_pad=np.int64(pad))
    sl = torch.tensor([10, 7, 5, 2])
    out.update(mask_sl=sl, mask_bool=ROP.sequence_mask(sl), mask_f64=ROP.sequence_mask(sl, dtype=float))
    xr = R(10, 4, 3)
    out.update(rev_x=xr, rev_out=ROP.reverse_sequences(xr, sl))

    # mu-law
    u = torch.linspace(-1, 1, 41)
    out.update(mulaw_u=u, mulaw_16=MuLawEncode(16)(u), mulaw_8=MuLawEncode(8)(u))

    # annealer traces (beta: 0->1 over 10; free nats: constant 5 then cosine over 7 to 0)
    a = CosineAnnealer(anneal_steps=10, constant_steps=0, start_value=0, end_value=1)
    b = CosineAnnealer(anneal_steps=7, constant_steps=5, start_value=2.0, end_value=0.0)
    out["anneal_beta"] = np.array([a.step() for _ in range(15)])
    out["anneal_fn"] = np.array([b.step() for _ in range(15)])
    save("functions.npz", **out)


def replay_eps(seed, Tp, B, z):
    """The scripted cell draws randn_like(mu) once per step (variational.py:141-152): replay that order."""
    torch.manual_seed(seed)
    return torch.stack([torch.randn(B, z) for _ in range(Tp)], 0)


def run_vrnn(model, x, x_sl, seed, beta, free_nats, stack, z):
    model.zero_grad()
    Tp = math.ceil(x.size(1) / stack)
    eps = replay_eps(seed, Tp, x.size(0), z)
    torch.manual_seed(seed)
    loss, metrics, o = model(x, x_sl, beta=beta, free_nats=free_nats)
    loss.backward()
    # the replayed noise must reproduce the sampled latents exactly, or the fixture is useless
    z_replay = o.z.detach()
    return loss, metrics, o, eps, z_replay


def gen_vrnn_small():
    torch.manual_seed(11)
    m = RM.VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, num_mix=10, num_bins=2**16)
    x, x_sl = O.synth_batch(3, 76, seed=5, ragged=True)  # T=76 -> T'=10 with 4 frames of stack padding
    x_sl = torch.tensor([76, 61, 40])
    x = x * (torch.arange(76).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays = {}
    for tag, (beta, fn) in {"a": (1.0, 2.0), "b": (0.3, 0.0)}.items():
        loss, metrics, o, eps, _ = run_vrnn(m, x, x_sl, 123, beta, fn, 8, 16)
        arrays.update({f"{tag}_loss": loss, f"{tag}_elbo": o.elbo, f"{tag}_log_prob": o.log_prob, f"{tag}_kl": o.kl, f"{tag}_z": o.z, f"{tag}_h_n": o.h_n})
        arrays[f"{tag}_eps"] = eps
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
    arrays.update(x=x, x_sl=x_sl)
    for k, v in m.state_dict().items():
        arrays[f"sd.{k}"] = v
    save("vrnn_small.npz", **arrays)


def gen_vrnn_full():
    """Full C2 dimensions (h=256, z=256, s=64) on a short batch; weights are NOT stored — they are reproduced from
    `torch.manual_seed(0)` + identical construction order, pinned by per-tensor checksums."""
    torch.manual_seed(0)
    m = RM.VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, num_mix=10, num_bins=2**16)
    arrays = {}
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    B, T = 4, 1280
    x, x_sl = O.synth_batch(B, T, seed=0, ragged=True)
    loss, metrics, o, eps, _ = run_vrnn(m, x, x_sl, 123, 1.0, 2.0, 64, 256)
    arrays.update(x_sl=x_sl, x_cks=np.array([x.double().sum().item(), x.double().abs().sum().item()]))
    arrays.update(eps_cks=np.array([eps.double().sum().item(), eps.double().abs().sum().item()]))
    arrays.update(loss=loss, elbo=o.elbo, log_prob=o.log_prob, kl=o.kl, h_n_cks=np.array([o.h_n.double().sum().item(), o.h_n.double().abs().sum().item()]))
    arrays["metric_names"] = np.array([mm.name for mm in metrics])
    arrays["metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    arrays["grad_norms"] = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    arrays["grad_names"] = np.array([k for k, _ in m.named_parameters()])
    # a few full gradient tensors (small ones) for element-wise checks
    for k in ["vrnn.likelihood.params.weight", "vrnn.likelihood.params.bias", "vrnn.vrnn_cell.gru_cell.bias_hh", "vrnn.encoder.2.bias"]:
        arrays[f"grad.{k}"] = dict(m.named_parameters())[k].grad
    save("vrnn_full.npz", **arrays)


def gen_srnn():
    """SRNNAudio: reduced size (full tensors, ragged, smoothing on/off, carried states) + full C3 dims by checksum."""
    arrays = {}
    x, _ = O.synth_batch(3, 76, seed=7)
    x_sl = torch.tensor([76, 53, 30])
    x = x * (torch.arange(76).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays.update(x=x, x_sl=x_sl)
    for tag, smoothing, beta, fn in (("sm", True, 1.0, 2.0), ("ns", False, 0.5, 0.0)):
        torch.manual_seed(31)
        m = RM.SRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, smoothing=smoothing)
        eps = replay_eps(321, 10, 3, 16)
        torch.manual_seed(321)
        loss, metrics, o = m(x, x_sl, beta=beta, free_nats=fn)
        loss.backward()
        arrays.update({f"{tag}_loss": loss, f"{tag}_elbo": o.elbo, f"{tag}_log_prob": o.log_prob, f"{tag}_kl": o.kl,
                       f"{tag}_z": o.z, f"{tag}_d_n": o.d_n, f"{tag}_z_n": o.z_n, f"{tag}_eps": eps})
        if smoothing:
            arrays[f"{tag}_a_n"] = o.a_n
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, v in m.state_dict().items():
            arrays[f"{tag}_sd.{k}"] = v
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
        if smoothing:  # second split with carried states d_0, a_0, z_0 (experiment_srnn_audio.py:261-269)
            m.zero_grad()
            eps2 = replay_eps(99, 10, 3, 16)
            torch.manual_seed(99)
            loss2, _, o2 = m(x, x_sl, beta=beta, free_nats=fn, d_0=o.d_n.detach(), a_0=o.a_n.detach(), z_0=o.z_n.detach())
            arrays.update(c_loss=loss2, c_elbo=o2.elbo, c_eps=eps2, c_z=o2.z)

    torch.manual_seed(0)
    m = RM.SRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, smoothing=True)
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    xf, xf_sl = O.synth_batch(4, 1280, seed=0, ragged=True)
    eps = replay_eps(123, 20, 4, 256)
    torch.manual_seed(123)
    loss, metrics, o = m(xf, xf_sl, beta=1.0, free_nats=2.0)
    loss.backward()
    arrays.update(f_x_sl=xf_sl, f_loss=loss, f_elbo=o.elbo, f_log_prob=o.log_prob, f_kl=o.kl)
    arrays["f_metric_names"] = np.array([mm.name for mm in metrics])
    arrays["f_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    arrays["f_grad_norms"] = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    arrays["f_grad_names"] = np.array([k for k, _ in m.named_parameters()])
    arrays["f_grad.srnn.a_backward_recurrent.bias_hh_l0"] = m.srnn.a_backward_recurrent.bias_hh_l0.grad.clone()
    arrays["f_grad.srnn.encoder.2.bias"] = m.srnn.encoder[2].bias.grad.clone()
    save("srnn.npz", **arrays)


def gen_wavenet():
    """WaveNet + DMoL: reduced size with full tensors; BASELINE config C5 dims (5x10, C=96) on a short batch by checksum."""
    arrays = {}
    torch.manual_seed(41)
    lik = DiscretizedLogisticMixtureDense(16, 1, num_mix=10, num_bins=2**16)
    m = RM.WaveNet(likelihood=lik, n_layers=3, n_stacks=2, res_channels=16, kernel_size=2, base_dilation=2, n_stack_frames=1)
    x, _ = O.synth_batch(3, 50, seed=8)
    x_sl = torch.tensor([50, 33, 20])
    x = x * (torch.arange(50).unsqueeze(0) < x_sl.unsqueeze(1))
    for tag, pad_rf in (("s", True), ("n", False)):
        m.zero_grad()
        xr = x.clone().requires_grad_(True)
        torch.manual_seed(1)
        loss, metrics, o = m(xr, x_sl, pad_receptive_field=pad_rf)
        loss.backward()
        arrays.update({f"{tag}_loss": loss, f"{tag}_log_prob": o.log_prob, f"{tag}_ll_twise": o.log_prob_twise, f"{tag}_dx": xr.grad})
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
    arrays.update(s_x=x, s_x_sl=x_sl, s_rf=np.int64(m.receptive_field))
    for k, v in m.state_dict().items():
        arrays[f"s_sd.{k}"] = v

    torch.manual_seed(0)
    lik = DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16)
    m = RM.WaveNet(likelihood=lik, n_layers=10, n_stacks=5, res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1)
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    xf, xf_sl = O.synth_batch(2, 1500, seed=0, ragged=True)
    torch.manual_seed(1)
    loss, metrics, o = m(xf, xf_sl)
    loss.backward()
    arrays.update(f_x_sl=xf_sl, f_loss=loss, f_log_prob=o.log_prob, f_rf=np.int64(m.receptive_field))
    arrays["f_metric_names"] = np.array([mm.name for mm in metrics])
    arrays["f_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    arrays["f_grad_norms"] = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    arrays["f_grad_names"] = np.array([k for k, _ in m.named_parameters()])
    arrays["f_grad.causal.conv.weight"] = m.causal.conv.weight.grad.clone()
    arrays["f_grad.res_stack.res_blocks.49.conv1x1rs.bias"] = m.res_stack.res_blocks[49].conv1x1rs.bias.grad.clone()
    arrays["f_grad.res_stack.res_blocks.7.conv.weight"] = m.res_stack.res_blocks[7].conv.weight.grad.clone()
    save("wavenet.npz", **arrays)


def gen_rssm():
    """Reference RSSMCell stepped over a short sequence in its three posterior modes; scalar test loss
    sum(z*wz) + sum(h*wh) + 0.7 * sum(KL) so that every gradient path (direct, recurrent, KL) is exercised."""
    from blvm.modules.rssm import RSSMCell

    arrays = {}
    T, B, Z, H, C, E = 6, 5, 16, 32, 48, 32
    g = torch.Generator().manual_seed(17)
    enc, ctx = torch.randn(T, B, E, generator=g), torch.randn(T, B, C, generator=g)
    z0, h0 = torch.randn(B, Z, generator=g) * 0.3, torch.randn(B, H, generator=g) * 0.3
    wz, wh = torch.randn(T, B, Z, generator=g), torch.randn(T, B, H, generator=g)
    arrays.update(enc=enc, ctx=ctx, z0=z0, h0=h0, wz=wz, wh=wh)
    for tag, kw, c_dim in (("plain", {}, C), ("res", dict(residual_posterior=True), C), ("prec", dict(precision_posterior=True), C),
                           ("top", dict(precision_posterior=True), 0)):
        torch.manual_seed(51)
        cell = RSSMCell(z_dim=Z, h_dim=H, c_dim=c_dim, e_dim=E, **kw)
        eps = replay_eps(77, T, B, Z)
        torch.manual_seed(77)
        z0r, h0r = z0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        encr, ctxr = enc.clone().requires_grad_(True), ctx[..., :c_dim].clone().requires_grad_(True)
        state, zs, hs, kls = (z0r, h0r), [], [], []
        for t in range(T):
            state, d = cell(encr[t], state, ctxr[t])
            zs.append(state[0]); hs.append(state[1])
            kls.append(RV.kl_divergence_gaussian(d.enc_mu, d.enc_sd, d.prior_mu, d.prior_sd))
        zs, hs, kl = torch.stack(zs), torch.stack(hs), torch.stack(kls)
        loss = (zs * wz).sum() + (hs * wh).sum() + 0.7 * kl.sum()
        loss.backward()
        arrays.update({f"{tag}_eps": eps, f"{tag}_zs": zs, f"{tag}_hs": hs, f"{tag}_kl": kl.sum((0, 2)), f"{tag}_loss": loss,
                       f"{tag}_d_enc": encr.grad, f"{tag}_d_z0": z0r.grad, f"{tag}_d_h0": h0r.grad})
        if c_dim:
            arrays[f"{tag}_d_ctx"] = ctxr.grad
        for k, v in cell.state_dict().items():
            arrays[f"{tag}_sd.{k}"] = v
        for k, p in cell.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
    save("rssm.npz", **arrays)


def replay_eps_levels(seed, T_levels, B, z_sizes):
    """CWVAE runs its levels top-down, one randn_like(mu) per step (clockwork_vae.py:265-279): replay that order."""
    torch.manual_seed(seed)
    eps = [None] * len(T_levels)
    for l in range(len(T_levels) - 1, -1, -1):
        eps[l] = torch.stack([torch.randn(B, z_sizes[l]) for _ in range(T_levels[l])], 0)
    return eps


def gen_cwvae():
    """CWVAEAudio: reduced size with full tensors (precision-weighted and residual posteriors, ragged lengths, free nats,
    carried state) and BASELINE config C4 dims (h=192, z=[128,64,32], strides [64,16,16], 8 blocks/level) on a short batch
    pinned by checksums."""
    arrays = {}
    cfg = dict(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
               num_mix=10, num_bins=2**16)
    B, T = 3, 150
    x, _ = O.synth_batch(B, T, seed=17)
    x_sl = torch.tensor([150, 97, 41])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays.update(x=x, x_sl=x_sl)
    for tag, kw, beta, fn in (("pw", dict(precision_posterior=True), 1.0, 0.5), ("rs", dict(residual_posterior=True), 0.7, 0.0)):
        torch.manual_seed(51)
        m = RM.CWVAEAudio(**cfg, **kw)
        # level lengths under same padding: ceil(previous / stride)
        T_l, n = [], T
        for s in cfg["strides"]:
            n = math.ceil(n / s)
            T_l.append(n)
        eps = replay_eps_levels(77, T_l, B, cfg["z_size"])
        torch.manual_seed(77)
        loss, metrics, o = m(x, x_sl, beta=beta, free_nats=fn)
        loss.backward()
        arrays.update({f"{tag}_loss": loss, f"{tag}_elbo": o.elbo, f"{tag}_log_prob": o.log_prob, f"{tag}_kld": o.kld,
                       f"{tag}_T_l": np.array(T_l)})
        for l in range(3):
            arrays.update({f"{tag}_eps{l}": eps[l], f"{tag}_z{l}": o.z[l], f"{tag}_enc_mu{l}": o.enc_mus[l],
                           f"{tag}_prior_mu{l}": o.prior_mus[l], f"{tag}_z_sl{l}": o.z_sl[l],
                           f"{tag}_state_z{l}": o.state_n[l][0], f"{tag}_state_h{l}": o.state_n[l][1]})
        arrays[f"{tag}_params"] = o.reconstructions_parameters[0]  # logits of the DMoL head, [B,T,1,10]
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, v in m.state_dict().items():
            arrays[f"{tag}_sd.{k}"] = v
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
        if tag == "pw":  # second call with the carried per-level state (experiment_clockwork_audio.py:263-271)
            m.zero_grad()
            eps2 = replay_eps_levels(5, T_l, B, cfg["z_size"])
            state0 = [(z.detach(), h.detach()) for z, h in o.state_n]
            torch.manual_seed(5)
            loss2, _, o2 = m(x, x_sl, state0=state0, beta=beta, free_nats=fn)
            arrays.update(c_loss=loss2, c_elbo=o2.elbo)
            for l in range(3):
                arrays[f"c_eps{l}"] = eps2[l]
                arrays[f"c_z{l}"] = o2.z[l]

    # BASELINE config C4 dimensions, short batch
    torch.manual_seed(0)
    full = dict(z_size=[128, 64, 32], h_size=192, strides=[64, 16, 16], num_level_layers=8, stride_per_layer=2,
                precision_posterior=True, likelihood="DMoL", num_bins=2**16)
    m = RM.CWVAEAudio(**full)
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    Bf, Tf = 2, 16384  # lengths where the (sic) same-padding arithmetic of clockwork_vae.py:245 is consistent
    xf, _ = O.synth_batch(Bf, Tf, seed=0)
    xf_sl = torch.tensor([16384, 12000])
    T_l, n = [], Tf
    for s in full["strides"]:
        n = math.ceil(n / s)
        T_l.append(n)
    eps = replay_eps_levels(123, T_l, Bf, full["z_size"])
    torch.manual_seed(123)
    loss, metrics, o = m(xf, xf_sl, beta=1.0, free_nats=4.0)
    loss.backward()
    arrays.update(f_x_sl=xf_sl, f_loss=loss, f_elbo=o.elbo, f_log_prob=o.log_prob, f_kld=o.kld, f_T_l=np.array(T_l))
    for l in range(3):
        arrays[f"f_z{l}"] = o.z[l]
    arrays["f_metric_names"] = np.array([mm.name for mm in metrics])
    arrays["f_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    # No gradients are stored at these dimensions: at random init the C4 network is chaotic in fp32 (the reference's own
    # fp32 activations differ by tens of percent from a float64 evaluation, and a single top-level step makes every
    # per-channel norm of the top decoder degenerate), so its gradients are not a usable fixture.  Gradient parity is
    # pinned on the reduced model above and per kernel at C4 widths (tests/test_gpu_convcoder.py, test_gpu_cwvae.py).
    arrays["f_grad_names"] = np.array([k for k, _ in m.named_parameters()])
    save("cwvae.npz", **arrays)


def gen_stcn():
    """STCN (top-down, dense, precision posterior, DMoL): reduced size with full tensors (stacked and single frames, ragged
    lengths, free nats) and the default configuration (25 blocks, C=256, latents [256..16], 64-frame stacks) by checksum."""
    from blvm.models import STCN

    arrays = {}
    for tag, S, T, beta, fn in (("s8", 8, 203, 1.0, 1.5), ("s1", 1, 61, 0.6, 0.0)):
        cfg = dict(likelihood="DMoL", n_layers=3, latent_size=[16, 16, 32], res_channels=16, n_stack_frames=S)
        torch.manual_seed(61)
        m = STCN(**cfg)
        B = 3
        x, _ = O.synth_batch(B, T, seed=19)
        x_sl = torch.tensor([T, int(T * 0.7), int(T * 0.3)])
        x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
        Tp = math.ceil(T / S)
        torch.manual_seed(88)  # one randn_like(mu) per level, top level first (stcn.py:309-325)
        eps = [None] * 3
        for l in (2, 1, 0):
            eps[l] = torch.randn(B, Tp, cfg["latent_size"][l])
        torch.manual_seed(88)
        loss, metrics, o = m(x, x_sl, beta=beta, free_nats=fn)
        loss.backward()
        arrays.update({f"{tag}_x": x, f"{tag}_x_sl": x_sl, f"{tag}_loss": loss, f"{tag}_elbo": o.elbo, f"{tag}_log_prob": o.log_prob})
        for l in range(3):
            arrays.update({f"{tag}_eps{l}": eps[l], f"{tag}_z{l}": o.z[l], f"{tag}_enc_mu{l}": o.enc_mus[l],
                           f"{tag}_prior_mu{l}": o.prior_mus[l], f"{tag}_kld{l}": o.klds[l]})
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, v in m.state_dict().items():
            arrays[f"{tag}_sd.{k}"] = v
        for k, p in m.named_parameters():
            if p.grad is not None:
                arrays[f"{tag}_grad.{k}"] = p.grad
        arrays[f"{tag}_nograd"] = np.array([k for k, p in m.named_parameters() if p.grad is None])

    torch.manual_seed(0)
    full = dict(likelihood="DMoL", n_layers=5, latent_size=[256, 128, 64, 32, 16], res_channels=256, n_stack_frames=64, dense=True)
    m = STCN(**full)
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    Bf, Tf = 4, 4000
    xf, xf_sl = O.synth_batch(Bf, Tf, seed=0, ragged=True)
    Tp = math.ceil(Tf / 64)
    torch.manual_seed(123)
    eps = [None] * 5
    for l in (4, 3, 2, 1, 0):
        eps[l] = torch.randn(Bf, Tp, full["latent_size"][l])
    torch.manual_seed(123)
    loss, metrics, o = m(xf, xf_sl, beta=1.0, free_nats=2.0)
    loss.backward()
    arrays.update(f_x_sl=xf_sl, f_loss=loss, f_elbo=o.elbo, f_log_prob=o.log_prob)
    for l in range(5):
        arrays[f"f_kld{l}"] = o.klds[l]
        arrays[f"f_z{l}"] = o.z[l]
    arrays["f_metric_names"] = np.array([mm.name for mm in metrics])
    arrays["f_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    gn = [(k, p.grad.double().norm().item()) for k, p in m.named_parameters() if p.grad is not None]
    arrays["f_grad_names"] = np.array([k for k, _ in gn])
    arrays["f_grad_norms"] = np.array([v for _, v in gn])
    for k in ("causal.conv.weight", "prior.0.transform_sd.4.weight", "out_transform.res_blocks.4.conv1x1rs.weight", "res_stack.res_blocks.12.conv.bias"):
        arrays[f"f_grad.{k}"] = dict(m.named_parameters())[k].grad.clone()
    save("stcn.npz", **arrays)


def gen_heads():
    """VRNNAudio with the Gaussian-mixture likelihood head (reduced size, ragged lengths): loss / ELBO / log-likelihood /
    latents and every parameter gradient.  (likelihood="Gaussian" raises in the reference for every model: its log_prob
    comes back [B,T,1] and is multiplied with a [B,T] mask, vrnn.py:268 — there is no reference output to record.)"""
    arrays = {}
    x, x_sl = O.synth_batch(3, 76, seed=5, ragged=True)
    arrays.update(x=x, x_sl=x_sl)
    for tag, lik in (("gmm", "GMM"),):
        torch.manual_seed(71)
        m = RM.VRNNAudio(likelihood=lik, input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, num_mix=10)
        eps = replay_eps(17, 10, 3, 16)
        torch.manual_seed(17)
        loss, metrics, o = m(x, x_sl, beta=0.8, free_nats=1.0)
        loss.backward()
        arrays.update({f"{tag}_loss": loss, f"{tag}_elbo": o.elbo, f"{tag}_log_prob": o.log_prob, f"{tag}_kl": o.kl, f"{tag}_z": o.z,
                       f"{tag}_eps": eps})
        for k, v in m.state_dict().items():
            arrays[f"{tag}_sd.{k}"] = v
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
    save("heads.npz", **arrays)


def gen_generate():
    """Ancestral sampling: CWVAEAudio.generate with the mode of the observation model (the latents are still sampled; the
    draw order — top level first, one randn(B, z) per step — is replayed into eps)."""
    arrays = {}
    cfg = dict(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
               num_mix=10, num_bins=2**16, precision_posterior=True)
    torch.manual_seed(51)
    m = RM.CWVAEAudio(**cfg)
    B, T = 2, 64
    torch.manual_seed(3)
    (x, x_sl), _ = m.generate(n_samples=B, max_timesteps=T, use_mode_observations=True)
    # Replay of the draws: level lengths are T // 16 at the top and, below, the length of the context the reference's own
    # decoder returns (its same-padding arithmetic in generate is the positional-argument variant, clockwork_vae.py:357).
    from blvm.utils.padding import get_same_padding

    eps, ctx_len, os_ = [None] * 3, None, [4, 8, 16]
    torch.manual_seed(3)
    with torch.no_grad():
        for l in (2, 1, 0):
            T_l = T // os_[l] if l == 2 else ctx_len
            eps[l] = torch.stack([torch.randn(B, cfg["z_size"][l]) for _ in range(T_l)], 0)
            length = math.ceil(T / cfg["strides"][l - 1]) if l > 0 else T
            pad = get_same_padding(length, m.cwvae.receptive_fields[l], cfg["strides"][l])
            ctx_len = m.cwvae.decoder[l](torch.zeros(B, cfg["z_size"][l] + cfg["h_size"], T_l), pad_right=pad)[1].shape[-1]
    arrays.update(cw_x_mode=x, cw_x_sl=x_sl, cw_T=np.array([T]))
    for l in range(3):
        arrays[f"cw_eps{l}"] = eps[l]
    for k, v in m.state_dict().items():
        arrays[f"cw_sd.{k}"] = v

    # VRNNAudio.generate, use_mode=True: observation modes are fed back, but `VRNN.generate` does not forward use_mode to the
    # cell (vrnn.py:405), so z is still SAMPLED from the prior: one randn(B, z) per step, replayed into eps
    torch.manual_seed(11)
    v = RM.VRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, num_mix=10, num_bins=2**16)
    arrays["vr_eps"] = replay_eps(29, 6, 3, 16)
    torch.manual_seed(29)
    (xv, xv_sl), _ = v.generate(n_samples=3, max_timesteps=6, use_mode=True)
    arrays.update(vr_x=xv, vr_x_sl=xv_sl)
    for k, p in v.state_dict().items():
        arrays[f"vr_sd.{k}"] = p

    # SRNNAudio.generate: per step one randn(B, z) (prior sample), then the head sampler's uniform_(1e-5, 1-1e-5) over the
    # logits' shape [B,S,K] and uniform_(1e-8, 1-1e-8) over [B,S,1] (variational.py:337,291) — replayed in that order
    torch.manual_seed(13)
    sr = RM.SRNNAudio(likelihood="DMoL", input_size=8, hidden_size=32, latent_size=16, residual_posterior=True, smoothing=True)
    Bn, Tn = 3, 5
    torch.manual_seed(41)
    e_l, u_l, u2_l = [], [], []
    for _ in range(Tn):
        e_l.append(torch.randn(Bn, 16))
        u_l.append(torch.empty(Bn, 8, 10).uniform_(1e-5, 1 - 1e-5))
        u2_l.append(torch.empty(Bn, 8, 1).uniform_(1e-8, 1 - 1e-8))
    torch.manual_seed(41)
    (xs_, xs_sl), _ = sr.generate(n_samples=Bn, max_timesteps=Tn)
    arrays.update(sr_x=xs_, sr_x_sl=xs_sl, sr_eps=torch.stack(e_l), sr_u=torch.stack(u_l), sr_u2=torch.stack(u2_l))
    for k, p in sr.state_dict().items():
        arrays[f"sr_sd.{k}"] = p

    # WaveNet.generate: per frame the head sampler draws uniform_(1e-5, 1-1e-5) over [B,1,K] and uniform_(1e-8, 1-1e-8) over [B,1,1]
    torch.manual_seed(15)
    wn = RM.WaveNet(likelihood=DiscretizedLogisticMixtureDense(16, 1, num_mix=10, num_bins=2**16), n_layers=3, n_stacks=2, res_channels=16)
    torch.manual_seed(43)
    wu, wu2 = [], []
    for _ in range(7):
        wu.append(torch.empty(2, 1, 10).uniform_(1e-5, 1 - 1e-5))
        wu2.append(torch.empty(2, 1, 1).uniform_(1e-8, 1 - 1e-8))
    torch.manual_seed(43)
    xw = wn.generate(n_samples=2, n_frames=7)
    arrays.update(wn_x=xw, wn_u=torch.stack(wu), wn_u2=torch.stack(wu2))
    for k, p in wn.state_dict().items():
        arrays[f"wn_sd.{k}"] = p
    save("generate.npz", **arrays)


def gen_wavenet_stacked():
    """WaveNet on frame stacks (the s=64 / s=256 lines of experiments/benchmarks.txt:7-8): n_stack_frames = 4 at reduced size, a
    length that is not a multiple of the stack, full tensors and every gradient."""
    arrays = {}
    torch.manual_seed(43)
    lik = DiscretizedLogisticMixtureDense(16, 1, num_mix=10, num_bins=2**16)
    m = RM.WaveNet(likelihood=lik, n_layers=3, n_stacks=2, res_channels=16, kernel_size=2, base_dilation=2, n_stack_frames=4)
    x, _ = O.synth_batch(3, 203, seed=9)
    x_sl = torch.tensor([203, 150, 81])
    x = x * (torch.arange(203).unsqueeze(0) < x_sl.unsqueeze(1))
    for tag, pad_rf in (("s", True), ("n", False)):
        m.zero_grad()
        xr = x.clone().requires_grad_(True)
        torch.manual_seed(1)
        loss, metrics, o = m(xr, x_sl, pad_receptive_field=pad_rf)
        loss.backward()
        arrays.update({f"{tag}_loss": loss, f"{tag}_log_prob": o.log_prob, f"{tag}_dx": xr.grad})
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        for k, p in m.named_parameters():
            arrays[f"{tag}_grad.{k}"] = p.grad
    arrays.update(x=x, x_sl=x_sl, rf=np.int64(m.receptive_field))
    for k, v in m.state_dict().items():
        arrays[f"sd.{k}"] = v
    # generation on frame stacks (wavenet.py:254-293 with n_stack_frames > 1: the head is evaluated once per stacked sample and the
    # new samples become the channels of the next input frame); per frame the sampler draws uniforms over [B,4,K] and [B,4,1]
    torch.manual_seed(47)
    gu, gu2 = [], []
    for _ in range(5):
        gu.append(torch.empty(2, 4, 10).uniform_(1e-5, 1 - 1e-5))
        gu2.append(torch.empty(2, 4, 1).uniform_(1e-8, 1 - 1e-8))
    torch.manual_seed(47)
    with torch.no_grad():
        xg = m.generate(n_samples=2, n_frames=5)
    arrays.update(gen_x=xg, gen_u=torch.stack(gu), gen_u2=torch.stack(gu2))
    save("wavenet_stacked.npz", **arrays)


def gen_generate16():
    """Ancestral sampling at widths the one-launch decoders accept (all of S, H, Z, R multiples of 16): VRNNAudio and SRNNAudio
    with SAMPLED observations.  Draw order per step (vrnn.py:405-417, srnn.py:366-392, variational.py:337,291): randn(B, z) for the
    prior sample, then the head sampler's uniform_(1e-5, 1-1e-5) over [B,S,K] and uniform_(1e-8, 1-1e-8) over [B,S,1]."""
    arrays = {}

    def replay(seed, T, B, Z, S):
        torch.manual_seed(seed)
        e, u, u2 = [], [], []
        for _ in range(T):
            e.append(torch.randn(B, Z))
            u.append(torch.empty(B, S, 10).uniform_(1e-5, 1 - 1e-5))
            u2.append(torch.empty(B, S, 1).uniform_(1e-8, 1 - 1e-8))
        return torch.stack(e), torch.stack(u), torch.stack(u2)

    S, H, Z, B, T = 16, 32, 16, 5, 7
    torch.manual_seed(17)
    v = RM.VRNNAudio(likelihood="DMoL", input_size=S, hidden_size=H, latent_size=Z, residual_posterior=True, num_mix=10, num_bins=2**16)
    e, u, u2 = replay(31, T, B, Z, S)
    x0 = (torch.rand(B, S, 1, generator=torch.Generator().manual_seed(5)) * 0.2 - 0.1)
    torch.manual_seed(31)
    (xv, xv_sl), _ = v.generate(n_samples=B, max_timesteps=T, x=x0)
    arrays.update(vr_x=xv, vr_x_sl=xv_sl, vr_x0=x0, vr_eps=e, vr_u=u, vr_u2=u2)
    for k, p in v.state_dict().items():
        arrays[f"vr_sd.{k}"] = p

    torch.manual_seed(19)
    sr = RM.SRNNAudio(likelihood="DMoL", input_size=S, hidden_size=H, latent_size=Z, residual_posterior=True, smoothing=True)
    e, u, u2 = replay(37, T, B, Z, S)
    torch.manual_seed(37)
    (xs_, xs_sl), out = sr.generate(n_samples=B, max_timesteps=T)
    arrays.update(sr_x=xs_, sr_x_sl=xs_sl, sr_eps=e, sr_u=u, sr_u2=u2, sr_h_p=out.h_p)
    for k, p in sr.state_dict().items():
        arrays[f"sr_sd.{k}"] = p
    save("generate16.npz", **arrays)


def gen_data():
    """Host-side data front-end: batches of the reference's length samplers for a seeded `random`, pools, padded collation."""
    import random

    from blvm.data.batchers import DynamicTensorBatcher
    from blvm.data.samplers import LengthEvalSampler, LengthTrainSampler

    arrays = {}
    g = torch.Generator().manual_seed(3)
    lengths = (torch.randint(800, 64000, (700,), generator=g) // 16 * 16).tolist()  # ties on purpose
    arrays["lengths"] = np.array(lengths)
    random.seed(7)
    s = LengthTrainSampler(lengths, batch_len=16000 * 20, min_pool_size=64, max_pool_difference=4000.0)
    arrays["train_pools_flat"] = np.array([i for p in s.pools for i in p])
    arrays["train_pool_sizes"] = np.array([len(p) for p in s.pools])
    for ep in range(2):
        bs = list(iter(s))
        arrays[f"train_ep{ep}_flat"] = np.array([i for b in bs for i in b])
        arrays[f"train_ep{ep}_sizes"] = np.array([len(b) for b in bs])
    random.seed(9)
    s = LengthTrainSampler(lengths, batch_len="2max", min_pool_size=128, num_batches=11, drop_last=False, longest_first=False)
    bs = list(iter(s))
    arrays["train_nb_flat"] = np.array([i for b in bs for i in b])
    arrays["train_nb_sizes"] = np.array([len(b) for b in bs])
    for tag, kw in (("len", dict(batch_len=16000 * 30)), ("size", dict(batch_size=32))):
        e = LengthEvalSampler(lengths, **kw)
        bs = list(iter(e))
        arrays[f"eval_{tag}_flat"] = np.array([int(i) for b in bs for i in b])
        arrays[f"eval_{tag}_sizes"] = np.array([len(b) for b in bs])
    xs = [torch.randn(n, generator=g) for n in (50, 17, 33, 50, 1)]
    out, sl = DynamicTensorBatcher().collate(xs)
    arrays.update(collate_in=np.concatenate([x.numpy() for x in xs]), collate_lens=np.array([len(x) for x in xs]), collate_out=out, collate_sl=sl)
    xs2 = [torch.randn(3, n, generator=g) for n in (7, 4, 9)]  # [channels, T], dynamic last dimension
    out2, sl2 = DynamicTensorBatcher(dim=-1, pad_value=-1.0).collate(xs2)
    arrays.update(collate2_in=np.concatenate([x.reshape(-1).numpy() for x in xs2]), collate2_out=out2, collate2_sl=sl2)
    save("data.npz", **arrays)


def gen_lstm():
    """LSTMAudio: reduced size with full tensors, and BASELINE config C1 ([8,4000], h=256, s=64) pinned by checksums."""
    arrays = {}
    torch.manual_seed(21)
    m = RM.LSTMAudio(stack_size=8, hidden_size=32, num_layers=1, num_mix=10, num_bins=2**16)
    x, _ = O.synth_batch(4, 83, seed=6)
    x_sl = torch.tensor([83, 70, 41, 17])  # descending (pack_padded_sequence enforce_sorted)
    x = x * (torch.arange(83).unsqueeze(0) < x_sl.unsqueeze(1))
    torch.manual_seed(5)
    loss, metrics, o = m(x, x_sl)
    loss.backward()
    arrays.update(s_x=x, s_x_sl=x_sl, s_loss=loss, s_ll=o.ll, s_z=o.z, s_hn=o.s_n[0], s_cn=o.s_n[1])
    arrays["s_metric_names"] = np.array([mm.name for mm in metrics])
    arrays["s_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    for k, v in m.state_dict().items():
        arrays[f"s_sd.{k}"] = v
    for k, p in m.named_parameters():
        arrays[f"s_grad.{k}"] = p.grad

    torch.manual_seed(0)
    m = RM.LSTMAudio(stack_size=64, hidden_size=256, num_layers=1, num_mix=10, num_bins=2**16)
    names = []
    for k, v in m.state_dict().items():
        names.append(k)
        arrays[f"cks.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item(), *v.shape], dtype=np.float64)
    arrays["param_names"] = np.array(names)
    for tag, ragged in (("full", False), ("ragged", True)):
        m.zero_grad()
        x, x_sl = O.synth_batch(8, 4000, seed=0, ragged=ragged)
        torch.manual_seed(5)
        loss, metrics, o = m(x, x_sl)
        loss.backward()
        arrays.update({f"{tag}_x_sl": x_sl, f"{tag}_loss": loss, f"{tag}_ll": o.ll})
        arrays[f"{tag}_metric_names"] = np.array([mm.name for mm in metrics])
        arrays[f"{tag}_metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
        arrays[f"{tag}_grad_norms"] = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
        arrays[f"{tag}_grad.lstm.bias_hh_l0"] = m.lstm.bias_hh_l0.grad.clone()
        arrays[f"{tag}_grad.likelihood.params.weight"] = m.likelihood.params.weight.grad.clone()
    arrays["grad_names"] = np.array([k for k, _ in m.named_parameters()])
    save("lstm.npz", **arrays)


def gen_lstm_layers():
    """LSTMAudio with num_layers = 2 (reduced size, ragged descending lengths): full tensors, carried states of both layers, every
    gradient; and a second call from the first call's states."""
    arrays = {}
    torch.manual_seed(23)
    m = RM.LSTMAudio(stack_size=8, hidden_size=32, num_layers=2, num_mix=10, num_bins=2**16)
    x, _ = O.synth_batch(4, 83, seed=6)
    x_sl = torch.tensor([83, 70, 41, 17])
    x = x * (torch.arange(83).unsqueeze(0) < x_sl.unsqueeze(1))
    loss, metrics, o = m(x, x_sl)
    loss.backward()
    arrays.update(x=x, x_sl=x_sl, loss=loss, ll=o.ll, z=o.z, hn=o.s_n[0], cn=o.s_n[1])
    for k, v in m.state_dict().items():
        arrays[f"sd.{k}"] = v
    for k, p in m.named_parameters():
        arrays[f"grad.{k}"] = p.grad
    with torch.no_grad():
        loss2, _, o2 = m(x, x_sl, s_0=(o.s_n[0].detach(), o.s_n[1].detach()))
    arrays.update(c_loss=loss2, c_ll=o2.ll)
    save("lstm_layers.npz", **arrays)


def gen_stcn_bottom_up():
    """STCN(top_down=False) (stcn.py:165-170, 284-287, 310-316): each latent conditions on the one BELOW it, levels are visited
    bottom level first, and the KL is the Monte-Carlo estimate log q(z) - log p(z) at the drawn z.  Reduced size, frame stacks,
    ragged lengths, free nats; full tensors and every gradient."""
    from blvm.models import STCN

    arrays = {}
    S, T, beta, fn = 8, 203, 0.8, 1.5
    cfg = dict(likelihood="DMoL", n_layers=3, latent_size=[16, 16, 32], res_channels=16, n_stack_frames=S, top_down=False)
    torch.manual_seed(61)
    m = STCN(**cfg)
    B = 3
    x, _ = O.synth_batch(B, T, seed=19)
    x_sl = torch.tensor([T, int(T * 0.7), int(T * 0.3)])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    Tp = math.ceil(T / S)
    torch.manual_seed(88)  # one randn_like(mu) per level, BOTTOM level first
    eps = [None] * 3
    for l in (0, 1, 2):
        eps[l] = torch.randn(B, Tp, cfg["latent_size"][l])
    torch.manual_seed(88)
    loss, metrics, o = m(x, x_sl, beta=beta, free_nats=fn)
    loss.backward()
    arrays.update(x=x, x_sl=x_sl, loss=loss, elbo=o.elbo, log_prob=o.log_prob)
    for l in range(3):
        arrays.update({f"eps{l}": eps[l], f"z{l}": o.z[l], f"enc_mu{l}": o.enc_mus[l], f"prior_mu{l}": o.prior_mus[l], f"kld{l}": o.klds[l]})
    arrays["metric_names"] = np.array([mm.name for mm in metrics])
    arrays["metric_values"] = np.array([mm.value for mm in metrics], dtype=np.float64)
    for k, v in m.state_dict().items():
        arrays[f"sd.{k}"] = v
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrays[f"grad.{k}"] = p.grad
    arrays["nograd"] = np.array([k for k, p in m.named_parameters() if p.grad is None])
    save("stcn_bottom_up.npz", **arrays)


def gen_cwvae_resets():
    """CWVAE(with_resets=True) (clockwork_vae.py:273-275: the state of every level below the top is reset to zeros whenever the
    level above ticks, including t = 0): the reduced CWVAEAudio of gen_cwvae (same seed and configuration, so the same weights)
    with the flag set on its CWVAE; ragged lengths, free nats, a level whose length is not a multiple of its parent's stride."""
    cfg = dict(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
               num_mix=10, num_bins=2**16)
    B, T = 3, 150
    x, _ = O.synth_batch(B, T, seed=17)
    x_sl = torch.tensor([150, 97, 41])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    torch.manual_seed(51)
    m = RM.CWVAEAudio(**cfg, precision_posterior=True)
    m.cwvae.with_resets = True
    T_l, n = [], T
    for s_ in cfg["strides"]:
        n = math.ceil(n / s_)
        T_l.append(n)
    eps = replay_eps_levels(77, T_l, B, cfg["z_size"])
    torch.manual_seed(77)
    loss, metrics, o = m(x, x_sl, beta=1.0, free_nats=0.5)
    loss.backward()
    arrays = dict(x=x, x_sl=x_sl, loss=loss, elbo=o.elbo, log_prob=o.log_prob, kld=o.kld, T_l=np.array(T_l))
    for l in range(3):
        arrays.update({f"eps{l}": eps[l], f"z{l}": o.z[l], f"state_z{l}": o.state_n[l][0], f"state_h{l}": o.state_n[l][1]})
    for k, v in m.state_dict().items():
        arrays[f"sd.{k}"] = v
    for k, p in m.named_parameters():
        arrays[f"grad.{k}"] = p.grad
    # generation with resets (clockwork_vae.py:369-371): replayed prior noise, the mode of the observation model
    from blvm.utils.padding import get_same_padding

    Tg, Bg = 64, 2
    torch.manual_seed(9)
    with torch.no_grad():
        (xg, _), _ = m.generate(n_samples=Bg, max_timesteps=Tg, use_mode_observations=True)
    geps, ctx_len, os_ = [None] * 3, None, [4, 8, 16]  # the prior draws of that call, replayed: top level first, one randn per step
    torch.manual_seed(9)
    with torch.no_grad():
        for l in (2, 1, 0):
            Tg_l = Tg // os_[l] if l == 2 else ctx_len
            geps[l] = torch.stack([torch.randn(Bg, cfg["z_size"][l]) for _ in range(Tg_l)], 0)
            length = math.ceil(Tg / cfg["strides"][l - 1]) if l > 0 else Tg
            pad = get_same_padding(length, m.cwvae.receptive_fields[l], cfg["strides"][l])
            ctx_len = m.cwvae.decoder[l](torch.zeros(Bg, cfg["z_size"][l] + cfg["h_size"], Tg_l), pad_right=pad)[1].shape[-1]
    arrays.update(gen_x=xg, gen_T=np.array([Tg]))
    for l in range(3):
        arrays[f"gen_eps{l}"] = geps[l]
    save("cwvae_resets.npz", **arrays)


def _merge_like_tracker(per_split_metrics):
    """What `tracker.update(metrics)` after every split leaves behind (tracker.py:377-392): the first metric of a name is copied,
    later ones merged into it with the metric's own rule.  Done with the REFERENCE's metric objects."""
    merged = {}
    for metrics in per_split_metrics:
        for mm in metrics:
            if mm.name in merged:
                merged[mm.name].update(mm)
            else:
                merged[mm.name] = mm.copy()
    names = list(merged)
    return np.array(names), np.array([merged[n].value for n in names], dtype=np.float64)


def gen_split_eval():
    """Split evaluation, the loops of the reference's entry points run on the reference's models:
    * WaveNet `split_sequence` + `forward_split` (wavenet.py:230-252, experiment_wavenet_audio.py:224-231) in both modes — "consume"
      (length > receptive field) and "extend" (length <= receptive field: every split left-padded to overlap + length), ragged batch,
      examples dropped as they end; per split: inputs, lengths, loss, log-prob, metrics; and the merged tracker values;
    * STCN `forward_split` (stcn.py:332-342; its `split_sequence` raises NotImplementedError, :328-330) for i_split = 0 and 1;
    * CWVAE `split_sequence` + `forward_split` (clockwork_vae.py:163-198, experiment_clockwork_audio.py:255-271): the splits the
      reference cuts, the single-split evaluation (is_last_split -> same padding), and what the reference does on every split that
      is not the last: `pad_same=False` raises IndexError for every configuration and length probed (the (sic) positional call of
      get_same_padding, SURVEY quirk 8, leaves x_sl unreduced, so the per-example stop index lies beyond the un-padded level
      length) — the fixture records the exception type over a sweep of lengths."""
    from blvm.models import STCN

    arrays = {}
    # ---- WaveNet ----
    torch.manual_seed(41)
    lik = DiscretizedLogisticMixtureDense(16, 1, num_mix=10, num_bins=2**16)
    m = RM.WaveNet(likelihood=lik, n_layers=3, n_stacks=2, res_channels=16, kernel_size=2, base_dilation=2, n_stack_frames=1)
    B, T = 3, 120
    x, _ = O.synth_batch(B, T, seed=8)
    x_sl = torch.tensor([120, 77, 31])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays.update(wn_x=x, wn_x_sl=x_sl, wn_rf=np.int64(m.receptive_field))
    for k, v in m.state_dict().items():
        arrays[f"wn_sd.{k}"] = v
    for tag, length in (("consume", 40), ("extend", 10)):
        splits_x, splits_x_sl = m.split_sequence(x, x_sl, length=length)
        per_split = []
        with torch.no_grad():
            for i, (xs, xs_sl) in enumerate(zip(splits_x, splits_x_sl)):
                torch.manual_seed(i)
                loss, metrics, o = m.forward_split(xs, xs_sl, i_split=i)
                per_split.append(metrics)
                arrays.update({f"wn_{tag}_x{i}": xs, f"wn_{tag}_x_sl{i}": xs_sl, f"wn_{tag}_loss{i}": loss, f"wn_{tag}_log_prob{i}": o.log_prob,
                               f"wn_{tag}_ll_twise{i}": o.log_prob_twise,
                               f"wn_{tag}_metric_values{i}": np.array([mm.value for mm in metrics], dtype=np.float64)})
        arrays[f"wn_{tag}_length"] = np.int64(length)
        arrays[f"wn_{tag}_n"] = np.int64(len(splits_x))
        arrays[f"wn_{tag}_metric_names"] = np.array([mm.name for mm in per_split[0]])
        arrays[f"wn_{tag}_merged_names"], arrays[f"wn_{tag}_merged_values"] = _merge_like_tracker(per_split)

    # ---- STCN ----
    cfg = dict(likelihood="DMoL", n_layers=3, latent_size=[16, 16, 32], res_channels=16, n_stack_frames=8)
    torch.manual_seed(61)
    s = STCN(**cfg)
    B, T = 3, 400
    x, _ = O.synth_batch(B, T, seed=19)
    x_sl = torch.tensor([400, 280, 120])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays.update(st_x=x, st_x_sl=x_sl, st_rf=np.int64(s.receptive_field))
    for k, v in s.state_dict().items():
        arrays[f"st_sd.{k}"] = v
    try:
        s.split_sequence(x, x_sl, length=100)
        arrays["st_split_sequence_raises"] = np.array("")
    except Exception as e:  # noqa: BLE001
        arrays["st_split_sequence_raises"] = np.array(type(e).__name__)
    for i in (0, 1):
        with torch.no_grad():
            torch.manual_seed(90 + i)
            loss, metrics, o = s.forward_split(x, x_sl, i_split=i, )
        Tp = o.z[0].shape[1]
        torch.manual_seed(90 + i)  # one randn_like(mu) per level, top level first (stcn.py:309-325)
        eps = [None] * 3
        for l in (2, 1, 0):
            eps[l] = torch.randn(B, Tp, cfg["latent_size"][l])
        arrays.update({f"st_loss{i}": loss, f"st_elbo{i}": o.elbo, f"st_log_prob{i}": o.log_prob,
                       f"st_metric_names{i}": np.array([mm.name for mm in metrics]),
                       f"st_metric_values{i}": np.array([mm.value for mm in metrics], dtype=np.float64)})
        for l in range(3):
            arrays.update({f"st_eps{i}_{l}": eps[l], f"st_z{i}_{l}": o.z[l], f"st_kld{i}_{l}": o.klds[l]})

    # ---- CW-VAE ----
    cfg = dict(z_size=[32, 16, 16], h_size=16, strides=[4, 2, 2], num_level_layers=2, stride_per_layer=2, likelihood="DMoL",
               num_mix=10, num_bins=2**16)
    torch.manual_seed(51)  # = the reduced model of gen_cwvae (same weights: cwvae.npz pw_sd.*)
    c = RM.CWVAEAudio(**cfg, precision_posterior=True)
    B, T = 3, 700
    x, _ = O.synth_batch(B, T, seed=17)
    x_sl = torch.tensor([700, 433, 150])
    x = x * (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1))
    arrays.update(cw_x=x, cw_x_sl=x_sl, cw_overall_rf=np.int64(c.cwvae.overall_receptive_field), cw_overall_stride=np.int64(c.cwvae.overall_stride))
    for length in (256, 300):
        sx, ssl = c.split_sequence(x, x_sl, length=length)
        arrays[f"cw_split{length}_shapes"] = np.array([list(t.shape) for t in sx])
        arrays[f"cw_split{length}_x_sl"] = np.stack([npy(t) for t in ssl])
        arrays[f"cw_split{length}_first"] = np.stack([npy(t[:, 0]) for t in sx])  # first sample of every split: pins the offsets
    # every split but the last: pad_same=False.  Sweep of lengths, full-length and ragged batches: the exception the reference raises
    raised = []
    with torch.no_grad():
        for L in list(range(157, 480, 7)) + [269, 301]:
            for sl in (torch.tensor([L, L, L]), torch.tensor([L, L - 40, 50])):
                try:
                    c.forward_split(x[:, :L], sl, is_last_split=False)
                    raised.append("")
                except Exception as e:  # noqa: BLE001
                    raised.append(type(e).__name__)
    arrays["cw_not_last_raises"] = np.array(sorted(set(raised)))
    arrays["cw_not_last_cases"] = np.int64(len(raised))
    # the evaluation loop when the utterances fit ONE split (the only case the reference's loop completes): is_last_split=True
    sx, ssl = c.split_sequence(x, x_sl, length=1024)
    assert len(sx) == 1
    T_l, n = [], sx[0].shape[1]
    for st in cfg["strides"]:
        n = math.ceil(n / st)
        T_l.append(n)
    eps = replay_eps_levels(31, T_l, B, cfg["z_size"])
    with torch.no_grad():
        torch.manual_seed(31)
        loss, metrics, o = c.forward_split(sx[0], ssl[0], state0=None, is_last_split=True)
    arrays.update(cw_one_x=sx[0], cw_one_x_sl=ssl[0], cw_one_loss=loss, cw_one_elbo=o.elbo, cw_one_log_prob=o.log_prob, cw_one_kld=o.kld)
    arrays["cw_one_merged_names"], arrays["cw_one_merged_values"] = _merge_like_tracker([metrics])
    for l in range(3):
        arrays.update({f"cw_one_eps{l}": eps[l], f"cw_one_z{l}": o.z[l], f"cw_one_state_z{l}": o.state_n[l][0], f"cw_one_state_h{l}": o.state_n[l][1]})
    save("split_eval.npz", **arrays)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["functions", "vrnn_small", "vrnn_full", "lstm", "srnn", "wavenet", "rssm", "cwvae", "stcn", "heads", "generate", "generate16", "wavenet_stacked", "cwvae_resets", "stcn_bottom_up", "lstm_layers", "data", "split_eval"]
    for w in which:
        globals()[f"gen_{w}"]()
