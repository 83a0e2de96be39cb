"""VRNN [Chung et al. 2015] with the reference's construction API and state_dict layout, computed by HIP kernels.

Reference: blvm/models/vrnn.py — `VRNNCell` :36-164, `VRNN` :167-434 (`compute_elbo` :255-279, `forward` :281-369),
`VRNNAudio` :437-545.  The modules below own the parameters in the same registration order (so seeded
initialisation and checkpoints are interchangeable with the reference) while `forward` hands whole sequences to
`libblvm_hip.so`: encoder/decoder MLPs (K6), the recurrent cell over all steps incl. BPTT (K1), the Gaussian KL with
free nats (K8, its backward fused into K1) and the DMoL head (K7).
"""
import math
from types import SimpleNamespace
from typing import Optional, Union

import torch
import torch.nn as nn
import torch.nn.init as init

from blvm import _hip, ops
from blvm.data.transforms import StackTensor
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, KLMetric, LatestMeanMetric, LLMetric, LossMetric
from blvm.models.base_model import BaseModel
from blvm.modules.convenience import View
from blvm.modules.distributions import (DiagonalGaussianDense, DiagonalGaussianMixtureDense,
                                        DiscretizedLogisticMixtureDense)  # fmt: skip

LIKELIHOOD_HEADS = (DiscretizedLogisticMixtureDense, DiagonalGaussianMixtureDense, DiagonalGaussianDense)


class LazyNamespace(SimpleNamespace):
    """SimpleNamespace whose expensive fields are computed on first access (the reference computes samples / modes /
    masks on every forward although the loss never uses them, vrnn.py:332-333).  A lazy callable that needs other fields
    takes the namespace as its argument (`lambda ns: ...`) instead of closing over it: a closure would make a reference cycle
    namespace -> dict -> lambda -> namespace that keeps the step's activations alive until the cyclic garbage collector runs."""

    def __init__(self, _lazy=None, **kwargs):
        super().__init__(**kwargs)
        object.__setattr__(self, "_lazy", dict(_lazy or {}))

    def __getattr__(self, name):
        lazy = object.__getattribute__(self, "_lazy")
        if name in lazy:
            fn = lazy.pop(name)
            value = fn(self) if fn.__code__.co_argcount else fn()
            setattr(self, name, value)
            return value
        raise AttributeError(name)


class VRNNCell(nn.Module):
    def __init__(self, x_dim: int, h_dim: int, z_dim: int, r_dim: Optional[int] = None, condition_h_on_x: bool = True,
                 residual_posterior: bool = False):  # fmt: skip
        super().__init__()
        r_dim = r_dim if r_dim else 2 * h_dim
        self.x_dim, self.h_dim, self.z_dim, self.r_dim = x_dim, h_dim, z_dim, r_dim
        self.condition_h_on_x = condition_h_on_x
        self.residual_posterior = residual_posterior

        def mlp(i, n):
            layers = []
            for k in range(n):
                layers += [nn.Linear(i if k == 0 else h_dim, h_dim), nn.ReLU()]
            return layers

        # registration (and therefore RNG) order as in the reference: phi_z, prior, posterior, gru_cell (vrnn.py:63-94)
        self.phi_z = nn.Sequential(*mlp(z_dim, 4))
        self.prior = nn.Sequential(*mlp(r_dim, 3), DiagonalGaussianDense(h_dim, z_dim))
        self.posterior = nn.Sequential(*mlp(x_dim + r_dim, 3), DiagonalGaussianDense(h_dim, z_dim))
        self.gru_cell = nn.GRUCell(x_dim + h_dim if condition_h_on_x else h_dim, r_dim)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init.orthogonal_(self.gru_cell.weight_hh)

    def get_initial_state(self, batch_size: int, device=None):
        device = device if device is not None else self.prior[0].weight.device
        return torch.zeros(batch_size, self.r_dim, device=device)

    def kernel_params(self):
        """Parameters in the order of `ops._VRNN_PARAM_ORDER` / struct BlvmVrnnWeights."""
        p, q, f, g = self.prior, self.posterior, self.phi_z, self.gru_cell
        out = []
        for seq in (p, q):
            for i in (0, 2, 4):
                out += [seq[i].weight, seq[i].bias]
            out += [seq[6].params.weight, seq[6].params.bias]
        for i in (0, 2, 4, 6):
            out += [f[i].weight, f[i].bias]
        out += [g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh]
        return out

    def sequence(self, enc, h0, eps, x_sl_dev, stride: int, free_nats: float = 0.0, generate: bool = False):
        """`generate=True`: z is drawn from the prior instead of the posterior (`VRNNCell.generate`, vrnn.py:143-164)."""
        if not self.condition_h_on_x:
            raise NotImplementedError("libblvm_hip: VRNN cell kernels implement condition_h_on_x=True (the VRNNAudio form)")
        head = self.prior[6]
        if head.initial_sd != 1 or self.posterior[6].epsilon != head.epsilon:
            raise NotImplementedError("libblvm_hip: Gaussian heads with initial_sd != 1 are not supported")
        mode = 3 if generate else int(self.residual_posterior)
        return ops.vrnn_sequence(enc, h0, eps, x_sl_dev, self.kernel_params(), self.x_dim, self.h_dim, self.z_dim,
                                 self.r_dim, mode, stride, free_nats, head.epsilon)  # fmt: skip

    @torch.no_grad()
    def generate(self, x: torch.Tensor, h: torch.Tensor, use_mode: bool = False, eps: Optional[torch.Tensor] = None):
        """One step of ancestral sampling (vrnn.py:143-164) as a length-1 sequence in prior-sampling mode."""
        B = x.size(0)
        if eps is None:
            eps = torch.zeros(B, self.z_dim, device=x.device) if use_mode else torch.randn(B, self.z_dim, device=x.device)
        x_sl = torch.ones(B, dtype=torch.int32, device=x.device)
        decin, _, _, _, _, mu_p, sd_p, z = self.sequence(x.unsqueeze(0).contiguous(), h, eps.view(1, B, self.z_dim).contiguous(),
                                                         x_sl, 1, 0.0, generate=True)  # fmt: skip
        h_new = decin[1, :, self.h_dim :]
        out = SimpleNamespace(h=h_new, z=z[0], enc_mu=mu_p[0], enc_sd=sd_p[0], prior_mu=mu_p[0], prior_sd=sd_p[0],
                              phi_z=decin[0, :, : self.h_dim])  # fmt: skip
        return h_new, out

    def forward(self, x: torch.Tensor, h: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """Single step (vrnn.py:109-141): a length-1 sequence through the same kernels."""
        B = x.size(0)
        eps = torch.randn(1, B, self.z_dim, device=x.device) if eps is None else eps.view(1, B, self.z_dim)
        x_sl = torch.ones(B, dtype=torch.int32, device=x.device)
        decin, _, _, mu_q, sd_q, mu_p, sd_p, z = self.sequence(x.unsqueeze(0).contiguous(), h, eps, x_sl, 1, 0.0)
        h_new = decin[1, :, self.h_dim :]
        out = SimpleNamespace(h=h_new, z=z[0], enc_mu=mu_q[0], enc_sd=sd_q[0], prior_mu=mu_p[0], prior_sd=sd_p[0],
                              phi_z=decin[0, :, : self.h_dim])  # fmt: skip
        return h_new, out


def _linears(seq: nn.Sequential):
    return [m for m in seq if isinstance(m, nn.Linear)]


class VRNN(nn.Module):
    def __init__(self, encoder: nn.Module, likelihood: nn.Module, x_dim: int, h_dim: int, z_dim: int,
                 r_dim: Optional[int] = None, decoder: nn.Module = None, residual_posterior: bool = False,
                 condition_h_on_x: bool = True, condition_x_on_h: bool = True, dropout: float = 0):  # fmt: skip
        super().__init__()
        r_dim = r_dim if r_dim else 2 * h_dim
        self.x_dim, self.h_dim, self.z_dim, self.r_dim = x_dim, h_dim, z_dim, r_dim
        self.residual_posterior = residual_posterior
        self.condition_h_on_x = condition_h_on_x
        self.condition_x_on_h = condition_x_on_h
        if dropout:
            raise NotImplementedError("libblvm_hip: dropout is not on the benchmark path (all runs use dropout 0)")

        self.encoder = encoder
        self.likelihood = likelihood
        if decoder is None:
            d_in = h_dim + r_dim if condition_x_on_h else h_dim
            self.decoder = nn.Sequential(nn.Linear(d_in, h_dim), nn.ReLU(), nn.Linear(h_dim, h_dim), nn.ReLU(),
                                         nn.Linear(h_dim, h_dim), nn.ReLU())  # fmt: skip
        else:
            self.decoder = decoder
        self.vrnn_cell = VRNNCell(x_dim=x_dim, h_dim=h_dim, z_dim=z_dim, condition_h_on_x=condition_h_on_x,
                                  residual_posterior=residual_posterior)  # fmt: skip
        self.dropout = None

    # ---- structure recognised by the fused path --------------------------------------------------------------------
    def _plan(self):
        enc, dec, lik = self.encoder, self.decoder, self.likelihood
        stack = next((m for m in enc if isinstance(m, StackTensor)), None) if isinstance(enc, nn.Sequential) else None
        ok = (
            stack is not None
            and isinstance(dec, nn.Sequential)
            and isinstance(lik, LIKELIHOOD_HEADS)
            and self.condition_h_on_x
            and self.condition_x_on_h
            and all(isinstance(m, (nn.Linear, nn.LeakyReLU, View, StackTensor)) for m in list(enc) + list(dec))
            and isinstance(dec[-2], nn.LeakyReLU)
        )
        if not ok:
            raise NotImplementedError(
                "libblvm_hip accelerates the VRNNAudio(likelihood='DMoL') structure (stacked-frame MLP encoder, MLP "
                "decoder, DMoL head, condition_h_on_x = condition_x_on_h = True); other compositions are not built yet"
            )
        return stack.n_frames, _linears(enc), _linears(dec), lik

    def forward(self, x: torch.Tensor, x_sl: torch.Tensor, beta: float = 1, free_nats: float = 0,
                h0: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None):  # fmt: skip
        """x [B,T] or [B,T,1] in [-1,1]; x_sl [B] lengths (host int tensor, as the reference's loaders give it).
        `eps` [T',B,z] optionally supplies the reparameterisation noise (otherwise drawn on the device)."""
        S, enc_lin, dec_lin, lik = self._plan()
        if x.ndim == 3:
            x = x.squeeze(-1)
        dev = x.device
        B, T = x.shape
        x_sl_host = x_sl.detach().cpu().to(torch.int64)
        x_sl_dev = ops.upload_i32(x_sl_host, dev)
        y = x.detach().to(torch.float32).contiguous()
        Tp = (T + S - 1) // S
        stride = math.ceil(T / Tp)
        H, Z, R = self.h_dim, self.z_dim, self.r_dim

        # frames stacked time-major: [T', B, S]   (StackTensor, operations.py:14-32: right zero-pad)
        xs = torch.nn.functional.pad(y, (0, Tp * S - T)) if Tp * S != T else y
        xs = xs.view(B, Tp, S).transpose(0, 1).contiguous().view(Tp * B, S)
        enc = ops.mlp(xs, enc_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE).view(Tp, B, -1)

        if eps is None:
            eps = torch.randn(Tp, B, Z, device=dev, dtype=torch.float32)
        decin, kld, kld_fn, mu_q, sd_q, mu_p, sd_p, z = self.vrnn_cell.sequence(enc, h0, eps, x_sl_dev, stride, free_nats)

        dec = ops.mlp(decin[:Tp].view(Tp * B, H + R), dec_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE)  # [T'*B, S*F]
        log_prob = lik.fused_log_prob(dec, y, x_sl_dev, ops.LAYOUT_TIME_MAJOR, B, T, Tp, S)  # K7 / K7b / K7c

        # ELBO assembly in float64 as the reference does (mask dtype `float`, vrnn.py:266-279)
        n_frames = float(x_sl_host.sum())
        elbo = log_prob - kld
        loss = -(log_prob - beta * kld_fn).sum() / n_frames
        kl = kld_fn  # the reference returns the free-nats-clamped KL as `kl` (vrnn.py:275-279); == kld when free_nats is 0

        sums = DeferredScalars(torch.stack([loss.detach(), elbo.detach().sum(), log_prob.detach().sum(), kl.detach().sum()]))
        ln2 = math.log(2)
        metrics = [
            LossMetric(sums[0], weight_by=B),
            LLMetric(sums[1], name="elbo", reduce_by=B),
            LLMetric(sums[2], name="rec", reduce_by=B),
            KLMetric(sums[3], reduce_by=B),
            KLMetric(sums[3] / ln2, name="kl (bpt)", reduce_by=n_frames),
            BitsPerDimMetric(sums[1], reduce_by=n_frames),
            LatestMeanMetric(beta, name="beta"),
            LatestMeanMetric(free_nats, name="free_nats"),
        ]

        max_len = int(x_sl_host.max())
        F = lik.out_features

        def parameters():
            d = dec.detach().view(Tp, B, S, F).permute(1, 0, 2, 3).reshape(B, Tp * S, F)[:, :max_len]
            return lik(d.contiguous())

        lazy = dict(
            reconstructions_parameters=parameters,
            reconstructions=lambda ns: lik.sample(ns.reconstructions_parameters),
            reconstructions_mode=lambda ns: lik.mode(ns.reconstructions_parameters),
            seq_mask=lambda: (torch.arange(max_len, device=dev).unsqueeze(0) < x_sl_dev.unsqueeze(1)).to(torch.float64),
        )
        outputs = LazyNamespace(
            lazy,
            elbo=elbo,
            log_prob=log_prob,
            kl=kl,
            y=y.unsqueeze(-1),
            z=z.transpose(0, 1),
            z_sl=(x_sl_host / stride).ceil().int(),
            h_n=decin[Tp - 1, :, H:],  # all_h[-1] after the pop at vrnn.py:310-311: the state ENTERING the last step
        )
        return loss, metrics, outputs


    @torch.no_grad()
    def generate(self, x: torch.Tensor, h0: Optional[torch.Tensor] = None, n_samples: int = 1, max_timesteps: int = 100,
                 stop_value: float = None, use_mode: bool = False, eps: Optional[torch.Tensor] = None, uniforms=None,
                 fused: Optional[bool] = None):  # fmt: skip
        """Autoregressive sampling (vrnn.py:371-434): the previous frame stack is encoded, the cell draws z from its prior and
        updates h, the decoder (on cat[phi_z, h_new] — the UPDATED state here, unlike `forward`) parameterises the next frame
        stack, which is sampled (or its mode taken) and fed back.  x [B,S,1] initial frame stack; returns ((x [B,1+T,S], x_sl), ns).
        `eps` [T,B,z] optionally supplies the prior noise, `uniforms` = (u [T,B,S,K], v [T,B,S]) the sampler's draws.  Every step
        runs the K6 / K1 / K7-head kernels at T' = 1; `fused=True` (DMoL head, no stop value) runs ALL steps in one launch (K1c);
        the default (None) takes the one-launch path whenever the model has that structure."""
        S, enc_lin, dec_lin, lik = self._plan()
        auto = fused is None
        if auto:
            c = self.vrnn_cell
            fused = (stop_value is None and max_timesteps > 0 and isinstance(lik, DiscretizedLogisticMixtureDense) and len(enc_lin) == 3
                     and len(dec_lin) == 3 and all(v % 16 == 0 for v in (S, c.h_dim, c.z_dim, c.r_dim)))  # fmt: skip
        if fused:
            try:
                return self._generate_fused(x, h0, n_samples, max_timesteps, stop_value, use_mode, eps, uniforms)
            except _hip.BlvmHipError:
                # the one-launch kernels have limits of their own beyond the structure test above (batch above the persistent
                # path's and shapes K1c's LDS plan does not take): only an EXPLICIT fused=True insists, the default serves every
                # shape the step-by-step path serves.  Nothing has run yet: the library validates before it launches.
                if not auto:
                    raise
        if x.size(0) > 1:
            assert x.size(0) == n_samples
        else:
            x = x.repeat(n_samples, *[1] * (x.ndim - 1))
        dev = x.device
        H = self.h_dim
        all_x = [x]
        x_sl = torch.ones(n_samples, dtype=torch.int)
        h = self.vrnn_cell.get_initial_state(n_samples, dev) if h0 is None else h0
        seq_active = torch.ones(n_samples, dtype=torch.int)
        t, all_ended = 0, False
        while not all_ended and t < max_timesteps:
            enc = ops.mlp(x.reshape(n_samples, S).to(torch.float32).contiguous(), enc_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE)
            # the reference does not forward use_mode to the cell (vrnn.py:405): z is always SAMPLED from the prior
            h, out = self.vrnn_cell.generate(enc, h.contiguous(), use_mode=False, eps=None if eps is None else eps[t])
            dec = ops.mlp(torch.cat([out.phi_z, h], -1).contiguous(), dec_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE)  # [B, S*F]
            parameters = lik(dec.view(n_samples, S, lik.out_features))
            if use_mode:
                x = lik.mode(parameters)
            else:
                x = lik.sample(parameters) if uniforms is None else lik.sample(parameters, uniforms=(uniforms[0][t], uniforms[1][t]))  # [B,S,1]
            all_x.append(x)
            x_sl += seq_active
            if stop_value is not None:
                ending = (x == stop_value).flatten(1).all(1).to(torch.int).cpu()
                seq_active *= 1 - ending
            t += 1
            all_ended = bool(torch.all(1 - seq_active))
        x = torch.cat(all_x, dim=-1).permute(0, 2, 1)
        return (x, x_sl), SimpleNamespace()


    @torch.no_grad()
    def _generate_fused(self, x, h0, n_samples, max_timesteps, stop_value, use_mode, eps, uniforms):
        S, enc_lin, dec_lin, lik = self._plan()
        cell = self.vrnn_cell
        if stop_value is not None or not isinstance(lik, DiscretizedLogisticMixtureDense) or len(enc_lin) != 3 or len(dec_lin) != 3:
            raise NotImplementedError("libblvm_hip: the one-launch decoder is built for the VRNNAudio(DMoL) structure without a stop value")
        if x.size(0) == 1:
            x = x.repeat(n_samples, *[1] * (x.ndim - 1))
        assert x.size(0) == n_samples
        dev, T, B = x.device, max_timesteps, n_samples
        if eps is None:
            eps = torch.randn(T, B, cell.z_dim, device=dev)
        if use_mode:
            u = v = None
        elif uniforms is None:
            u = torch.empty(T, B, S, lik.num_mix, device=dev).uniform_(1e-5, 1.0 - 1e-5)
            v = torch.empty(T, B, S, device=dev).uniform_(1e-8, 1.0 - 1e-8)
        else:
            u, v = uniforms[0].reshape(T, B, S, lik.num_mix).to(dev), uniforms[1].reshape(T, B, S).to(dev)
        slope = next(m.negative_slope for m in self.encoder if isinstance(m, nn.LeakyReLU))
        xs, _ = ops.vrnn_decode(enc_lin, cell.kernel_params(), dec_lin, lik.params, x.reshape(B, S), h0, eps[:T], u, v, S, cell.h_dim,
                                cell.z_dim, cell.r_dim, lik.num_mix, cell.prior[6].epsilon, slope, lik.log_epsilon)  # fmt: skip
        out = torch.cat([x.reshape(B, 1, S), xs], 1)  # [B,1+T,S] like the step-by-step path
        return (out, torch.full((B,), T + 1, dtype=torch.int)), SimpleNamespace()


class VRNNAudio(BaseModel):
    def __init__(self, likelihood: Union[str, nn.Module], input_size: int = 200, hidden_size: int = 256,
                 latent_size: int = 64, residual_posterior: bool = False, condition_h_on_x: bool = True,
                 condition_x_on_h: bool = True, num_mix: int = 10, num_bins: int = 256):  # fmt: skip
        super().__init__()
        self.likelihood = likelihood
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.latent_size = latent_size
        self.residual_posterior = residual_posterior
        self.condition_h_on_x = condition_h_on_x
        self.condition_x_on_h = condition_x_on_h
        self.num_mix = num_mix
        self.num_bins = num_bins

        if likelihood == "DMoL":
            # the reference hard-codes num_mix=10, num_bins=2**16 on this branch whatever the ctor args say
            # (vrnn.py:464-469); x_dim still follows num_mix.
            likelihood_module = DiscretizedLogisticMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=10, num_bins=2**16)
        elif likelihood == "GMM":
            likelihood_module = DiagonalGaussianMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=num_mix, initial_sd=1,
                                                             epsilon=1e-4)  # fmt: skip
        elif likelihood == "Gaussian":
            likelihood_module = DiagonalGaussianDense(x_dim=2, y_dim=1, epsilon=1e-4)
        else:
            raise ValueError(f"Unknown likelihood type {likelihood}")

        encoder = nn.Sequential(
            View(-1),
            StackTensor(input_size, dim=1),
            nn.Linear(input_size, hidden_size),
            nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size),
            nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size),
            nn.LeakyReLU(),
        )
        decoder = nn.Sequential(
            nn.Linear(3 * hidden_size, hidden_size),
            nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size),
            nn.LeakyReLU(),
            nn.Linear(hidden_size, input_size * likelihood_module.out_features),
            nn.LeakyReLU(),
            View(-1, likelihood_module.out_features),
        )
        self.vrnn = VRNN(encoder=encoder, decoder=decoder, likelihood=likelihood_module, x_dim=hidden_size, h_dim=hidden_size,
                         z_dim=latent_size, residual_posterior=residual_posterior, condition_h_on_x=condition_h_on_x,
                         condition_x_on_h=condition_x_on_h)  # fmt: skip

    def forward(self, x, x_sl, beta: float = 1, free_nats: float = 0, h0=None, eps=None):
        return self.vrnn(x, x_sl, beta, free_nats, h0, eps)

    def generate(self, n_samples: int = 1, max_timesteps: int = 100, use_mode: bool = False, x=None, h0=None, eps=None, uniforms=None,
                 fused: Optional[bool] = None):  # fmt: skip
        """Same arguments as the reference (vrnn.py:529-546); `fused`: every step in one launch (K1c; default: when it applies)."""
        x = torch.zeros(n_samples, self.input_size, 1, device=self.device) if x is None else x
        return self.vrnn.generate(n_samples=n_samples, max_timesteps=max_timesteps, stop_value=None, use_mode=use_mode, x=x, h0=h0,
                                  eps=eps, uniforms=uniforms, fused=fused)  # fmt: skip
