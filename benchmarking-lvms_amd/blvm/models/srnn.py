"""SRNN [Fraccaro et al. 2016] with the reference's construction API and state_dict layout, computed by HIP kernels.

Reference: blvm/models/srnn.py — `SRNN` :28-403 (`compute_elbo` :137-160, `forward` :162-302), `SRNNAudio` :406-536.
Pipeline: encoder MLP (K6) -> forward GRU d over the shifted encoding (K2) -> time-reversed GRU a over cat[x, d] (K2,
per-row reversal folded into the kernel) -> latent chain z_t | z_{t-1}, d_t, a_t (K3) -> decoder MLP (K6) -> DMoL (K7),
KL + free nats (K8, backward fused into K3).  Faithful quirk: `kl` in the outputs/metrics is the RAW KL while the
loss uses the free-nats-clamped one (srnn.py:154-160; the VRNN returns the clamped one).
"""
import math
from types import SimpleNamespace
from typing import Optional, Union

import torch
import torch.nn as nn
import torch.nn.init as init

from blvm import ops
from blvm.data.transforms import StackTensor
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, KLMetric, LatestMeanMetric, LLMetric, LossMetric
from blvm.models.base_model import BaseModel
from blvm.models.vrnn import LIKELIHOOD_HEADS, LazyNamespace, _linears
from blvm.modules.convenience import View
from blvm.modules.distributions import DiagonalGaussianDense, DiagonalGaussianMixtureDense, DiscretizedLogisticMixtureDense
from blvm.utils.operations import split_sequence
from blvm.utils.padding import get_modulo_length


class SRNN(nn.Module):
    def __init__(self, encoder, decoder, likelihood, x_dim, h_dim, z_dim, r_dim: Optional[int] = None,
                 gated_stochastic_transfer: bool = False, use_phi_z: bool = False, dropout: float = 0, num_layers: int = 1,
                 residual_posterior: bool = False, smoothing: bool = True):  # fmt: skip
        super().__init__()
        r_dim = 2 * h_dim if r_dim is None else r_dim
        if gated_stochastic_transfer or use_phi_z or dropout or num_layers != 1:
            # num_layers > 1 crashes in the reference itself (srnn.py:197); the other switches are never
            # set by SRNNAudio (srnn.py:475-485)
            raise NotImplementedError("libblvm_hip: SRNN is built in the form SRNNAudio constructs "
                                      "(Elman stochastic transfer, no phi_z, no dropout, one GRU layer)")
        self.encoder, self.decoder = encoder, decoder
        self.x_dim, self.h_dim, self.z_dim, self.r_dim = x_dim, h_dim, z_dim, r_dim
        self.use_phi_z, self.gated_stochastic_transfer = use_phi_z, gated_stochastic_transfer
        self.dropout, self.num_layers = None, num_layers
        self.residual_posterior, self.smoothing = residual_posterior, smoothing
        self.phi_z = None

        def mlp_head(i):
            return nn.Sequential(nn.Linear(i, h_dim), nn.LeakyReLU(), nn.Linear(h_dim, h_dim), nn.LeakyReLU(),
                                 nn.Linear(h_dim, h_dim), nn.LeakyReLU(), DiagonalGaussianDense(h_dim, z_dim))  # fmt: skip

        # registration / RNG order of the reference: posterior, prior, d-GRU, a-GRU (srnn.py:92-121)
        self.posterior = mlp_head(r_dim + z_dim)
        self.prior = mlp_head(r_dim + z_dim)
        self.d_forward_recurrent = nn.GRU(x_dim, r_dim, num_layers)
        if smoothing:
            self.a_backward_recurrent = nn.GRU(x_dim + r_dim, r_dim, num_layers)
        else:
            self.a_mlp = nn.Sequential(nn.Linear(x_dim + r_dim, r_dim), nn.LeakyReLU(), nn.Linear(r_dim, r_dim), nn.LeakyReLU())
        self.likelihood = likelihood
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init.orthogonal_(self.d_forward_recurrent.weight_hh_l0)
        if self.smoothing:
            init.orthogonal_(self.a_backward_recurrent.weight_hh_l0)

    def _chain_params(self):
        out = []
        for seq in (self.prior, self.posterior):
            for i in (0, 2, 4):
                out += [seq[i].weight, seq[i].bias]
            out += [seq[6].params.weight, seq[6].params.bias]
        return out

    def _plan(self):
        enc, dec, lik = self.encoder, self.decoder, self.likelihood
        stack = next((m for m in enc if isinstance(m, StackTensor)), None) if isinstance(enc, nn.Sequential) else None
        ok = (
            stack is not None
            and isinstance(dec, nn.Sequential)
            and isinstance(lik, LIKELIHOOD_HEADS)
            and all(isinstance(m, (nn.Linear, nn.LeakyReLU, View, StackTensor)) for m in list(enc) + list(dec))
            and isinstance(dec[-2], nn.LeakyReLU)
        )
        if not ok:
            raise NotImplementedError("libblvm_hip accelerates the SRNNAudio(likelihood='DMoL') structure")
        return stack.n_frames, _linears(enc), _linears(dec), lik

    def forward(self, x, x_sl, u=None, d_0=None, a_0=None, z_0=None, h_p_0=None, h_q_0=None, beta: float = 1,
                free_nats: float = 0, eps: Optional[torch.Tensor] = None):  # fmt: skip
        if u is not None:
            raise NotImplementedError("libblvm_hip: external control input u is not supported (SRNNAudio never passes it)")
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
        x_sl_strided = (x_sl_host / stride).ceil().int()
        lens_dev = ops.upload_i32(x_sl_strided, dev)
        H, Z, R = self.h_dim, self.z_dim, self.r_dim

        xs = torch.nn.functional.pad(y, (0, Tp * S - T)) if Tp * S != T else y
        xs = xs.view(B, Tp, S).transpose(0, 1).contiguous().view(Tp * B, S)
        enc = ops.mlp(xs, enc_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE).view(Tp, B, -1)

        # u_t = x_{t-1};  d = GRU(u);  d <- [d_0, d[:-1]]   (srnn.py:192-197)
        u_enc = torch.cat([torch.zeros_like(enc[:1]), enc[:-1]], 0)
        gd = self.d_forward_recurrent
        d0 = d_0.reshape(B, R) if d_0 is not None else None
        d_seq, d_n = ops.gru_sequence(u_enc, d0, gd.weight_ih_l0, gd.weight_hh_l0, gd.bias_ih_l0, gd.bias_hh_l0)
        d_first = d0.unsqueeze(0) if d0 is not None else torch.zeros(1, B, R, device=dev)
        d = torch.cat([d_first, d_seq[:-1]], 0)

        cat_xd = torch.cat([enc, d], -1)
        if self.smoothing:
            ga = self.a_backward_recurrent
            a0 = a_0.reshape(B, R) if a_0 is not None else None
            a, a_n = ops.gru_sequence(cat_xd, a0, ga.weight_ih_l0, ga.weight_hh_l0, ga.bias_ih_l0, ga.bias_hh_l0, lens_dev, True)
            a_n = a_n.unsqueeze(0)
        else:
            a = ops.mlp(cat_xd.view(Tp * B, -1), _linears(self.a_mlp), ops.ACT_LEAKY, ops.LEAKY_SLOPE).view(Tp, B, R)
            a_n = None

        if eps is None:
            eps = torch.randn(Tp, B, Z, device=dev, dtype=torch.float32)
        head = self.prior[6]
        zs, kld, kld_fn, mu_q, sd_q, mu_p, sd_p = ops.srnn_latent_chain(
            d, a, z_0, eps, x_sl_dev, self._chain_params(), H, Z, R, self.residual_posterior, stride, free_nats, head.epsilon
        )
        z = zs[1:]
        dec = ops.mlp(torch.cat([z, d], -1).view(Tp * B, Z + R), dec_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE)
        log_prob = lik.fused_log_prob(dec, y, x_sl_dev, ops.LAYOUT_TIME_MAJOR, B, T, Tp, S)  # K7 / K7b / K7c

        n_frames = float(x_sl_host.sum())
        elbo = log_prob - kld
        loss = -(log_prob - beta * kld_fn).sum() / n_frames
        kl = kld  # raw KL (srnn.py:156-160)

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
            p = dec.detach().view(Tp, B, S, F).permute(1, 0, 2, 3).reshape(B, Tp * S, F)[:, :max_len]
            return lik(p.contiguous())

        lazy = dict(
            parameters=parameters,
            reconstructions_parameters=lambda ns: ns.parameters,
            reconstructions=lambda ns: lik.sample(ns.parameters),
            reconstructions_mode=lambda ns: lik.mode(ns.parameters),
            seq_mask=lambda: (torch.arange(max_len, device=dev).unsqueeze(0) < x_sl_dev.unsqueeze(1)).to(torch.float64),
        )
        outputs = LazyNamespace(
            lazy, elbo=elbo, log_prob=log_prob, kl=kl, y=y.unsqueeze(-1), z=z.transpose(0, 1), z_sl=x_sl_strided,
            d_n=d_n.unsqueeze(0), a_n=a_n, z_n=z[-1], h_p_n=None, h_q_n=None,
        )  # fmt: skip
        return loss, metrics, outputs


    @torch.no_grad()
    def generate(self, x, u=None, d_0=None, a_0=None, z_0=None, h_p_0=None, n_samples: int = 1, max_timesteps: int = 100,
                 stop_value: float = None, use_mode: bool = False, eps=None, uniforms=None, fused: Optional[bool] = None):  # fmt: skip
        """Unconditional autoregressive sampling (srnn.py:304-403): encode the previous frame stack, one GRU step for d_t, draw
        z_t from the prior given cat[d_t, z_{t-1}] (its mean if use_mode), decode cat[z_t, d_t], SAMPLE the next frame stack and
        feed it back.  x [B,1,S] start frames.  Returns ((x [B,T,S,1], x_sl), ns(h_p)).  `eps` [T,B,z] and `uniforms`
        (list of the head sampler's draws per step) optionally supply the randomness.  Every step runs K6 / K2 / K3 at T' = 1;
        `fused` (DMoL head, no stop value, at most `blvm_pchain_max_batch()` utterances; default: whenever that holds) runs ALL steps
        in one persistent launch (K3c, `ops.srnn_generate`)."""
        if u is not None or x.size(1) > 1:
            raise NotImplementedError("libblvm_hip: SRNN.generate is built for unconditional generation (x [B,1,S], u=None)")
        S, enc_lin, dec_lin, lik = self._plan()
        dev = x.device
        H, Z, R = self.h_dim, self.z_dim, self.r_dim
        n = n_samples
        can_fuse = (stop_value is None and isinstance(lik, DiscretizedLogisticMixtureDense) and len(enc_lin) == 3 and len(dec_lin) == 3
                    and 0 < n <= ops.load().blvm_pchain_max_batch() and all(v % 16 == 0 for v in (S, H, Z, R)))  # fmt: skip
        if fused is None:
            fused = can_fuse
        if fused:
            if not can_fuse:
                raise NotImplementedError("libblvm_hip: the one-launch SRNN decoder needs the SRNNAudio(DMoL) structure, no stop value, "
                                          "dimensions in multiples of 16 and at most blvm_pchain_max_batch() utterances")  # fmt: skip
            return self._generate_fused(x, d_0, z_0, n, max_timesteps, use_mode, eps, uniforms, S, enc_lin, dec_lin, lik)
        x_sl = torch.zeros(n)
        d_t = torch.zeros(n, R, device=dev) if d_0 is None else d_0.reshape(n, R).contiguous()
        z_t = torch.zeros(n, Z, device=dev) if z_0 is None else z_0.contiguous()
        ones = torch.ones(n, dtype=torch.int32, device=dev)
        gd, head = self.d_forward_recurrent, self.prior[6]
        all_x = []
        seq_active = torch.ones(n, dtype=torch.int)
        t, all_ended = 0, False
        h_p = None
        while not all_ended and t < max_timesteps:
            enc = ops.mlp(x.reshape(n, S).to(torch.float32).contiguous(), enc_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE).view(1, n, -1)
            d_seq, _ = ops.gru_sequence(enc, d_t, gd.weight_ih_l0, gd.weight_hh_l0, gd.bias_ih_l0, gd.bias_hh_l0)
            d_t = d_seq[0].contiguous()
            h_p = torch.cat([d_t, z_t], -1)
            e = torch.zeros(1, n, Z, device=dev) if use_mode else (torch.randn(1, n, Z, device=dev) if eps is None else eps[t].view(1, n, Z))
            zs, *_ = ops.srnn_latent_chain(d_t.unsqueeze(0), torch.zeros(1, n, R, device=dev), z_t, e.to(dev).contiguous(), ones,
                                           self._chain_params(), H, Z, R, 3, 1, 0.0, head.epsilon)  # mode 3: z ~ prior
            z_t = zs[1].contiguous()
            dec = ops.mlp(torch.cat([z_t, d_t], -1).contiguous(), dec_lin, ops.ACT_LEAKY, ops.LEAKY_SLOPE)
            parameters = lik(dec.view(n, S, lik.out_features))
            xs = lik.sample(parameters) if uniforms is None else lik.sample(parameters, uniforms=uniforms[t])  # [B,S,1]
            all_x.append(xs)
            x = xs.unsqueeze(1)
            x_sl += seq_active
            if stop_value is not None:
                seq_active *= 1 - (xs == stop_value).flatten(1).all(1).to(torch.int).cpu()
            t += 1
            all_ended = bool(torch.all(1 - seq_active))
        return (torch.stack(all_x, dim=1), x_sl), SimpleNamespace(h_p=h_p)


    def _generate_fused(self, x, d_0, z_0, n, T, use_mode, eps, uniforms, S, enc_lin, dec_lin, lik):
        dev = x.device
        H, Z, R = self.h_dim, self.z_dim, self.r_dim
        if use_mode:  # the reference's use_mode takes the PRIOR's mean and still samples the observation (srnn.py:366-392)
            eps = torch.zeros(T, n, Z, device=dev)
        elif eps is None:
            eps = torch.randn(T, n, Z, device=dev)
        else:
            eps = torch.as_tensor(eps)[:T].reshape(T, n, Z).to(dev)
        if uniforms is None:
            u = torch.empty(T, n, S, lik.num_mix, device=dev).uniform_(1e-5, 1.0 - 1e-5)
            v = torch.empty(T, n, S, device=dev).uniform_(1e-8, 1.0 - 1e-8)
        else:
            u = torch.stack([uniforms[t][0].reshape(n, S, lik.num_mix) for t in range(T)]).to(dev)
            v = torch.stack([uniforms[t][1].reshape(n, S) for t in range(T)]).to(dev)
        slope = next(m.negative_slope for m in self.encoder if isinstance(m, nn.LeakyReLU))
        d0 = None if d_0 is None else d_0.reshape(n, R)
        xs, d_n, zs = ops.srnn_generate(enc_lin, self.d_forward_recurrent, self._chain_params(), dec_lin, lik.params, x.reshape(n, S), d0, z_0,
                                        eps, u, v, S, H, Z, R, lik.num_mix, self.prior[6].epsilon, slope, lik.log_epsilon)  # fmt: skip
        z_prev = zs[T - 2] if T > 1 else (torch.zeros(n, Z, device=dev) if z_0 is None else z_0)
        x_sl = torch.zeros(n) + T
        return (xs.unsqueeze(-1), x_sl), SimpleNamespace(h_p=torch.cat([d_n, z_prev], -1))


class SRNNAudio(BaseModel):
    def __init__(self, likelihood: Union[str, nn.Module], input_size: int = 200, hidden_size: int = 256, latent_size: int = 64,
                 dropout: float = 0, residual_posterior: bool = False, smoothing: bool = True, num_mix: int = 10,
                 num_bins: int = 256):  # fmt: skip
        super().__init__()
        self.likelihood = likelihood
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.latent_size = latent_size
        self.dropout = dropout
        self.residual_posterior = residual_posterior
        self.num_mix = num_mix
        self.num_bins = num_bins
        self.smoothing = smoothing

        if likelihood == "DMoL":
            # hard-coded num_mix / num_bins on this branch, as in the reference (srnn.py:433-438)
            likelihood_module = DiscretizedLogisticMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=10, num_bins=2**16)
        elif likelihood == "GMM":
            likelihood_module = DiagonalGaussianMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=num_mix, initial_sd=1,
                                                             epsilon=1e-4)  # fmt: skip
        elif likelihood == "Gaussian":
            likelihood_module = DiagonalGaussianDense(x_dim=2, y_dim=1, epsilon=1e-4)
        else:
            raise ValueError(f"Unknown likelihood type {likelihood}")

        encoder = nn.Sequential(
            View(-1), StackTensor(input_size, dim=1),
            nn.Linear(input_size, hidden_size), nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size), nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size), nn.LeakyReLU(),
        )  # fmt: skip
        decoder = nn.Sequential(
            nn.Linear(2 * hidden_size + latent_size, hidden_size), nn.LeakyReLU(),
            nn.Linear(hidden_size, hidden_size), nn.LeakyReLU(),
            nn.Linear(hidden_size, input_size * likelihood_module.out_features), nn.LeakyReLU(),
            View(-1, likelihood_module.out_features),
        )  # fmt: skip
        self.srnn = SRNN(encoder=encoder, decoder=decoder, likelihood=likelihood_module, x_dim=hidden_size, h_dim=hidden_size,
                         z_dim=latent_size, dropout=dropout, residual_posterior=residual_posterior, smoothing=smoothing)  # fmt: skip
        self.forward_split = self.forward

    def split_sequence(self, x, x_sl, length: int, drop_inactive: bool = False):
        """Split long sequences into stack-aligned sub-sequences without overlap (srnn.py:489-499); states d_n, a_n, z_n
        of one split are passed as d_0, a_0, z_0 of the next."""
        length = get_modulo_length(length, self.input_size, kernel_size=self.input_size)
        return split_sequence(x, x_sl, length=length, overlap=0, drop_inactive=drop_inactive)

    def forward(self, x, x_sl, beta: float = 1, free_nats: float = 0, d_0=None, a_0=None, z_0=None, eps=None):
        loss, metrics, outputs = self.srnn(x=x, x_sl=x_sl, d_0=d_0, a_0=a_0, z_0=z_0, beta=beta, free_nats=free_nats, eps=eps)
        outputs._lazy["x_hat"] = lambda ns: self.srnn.likelihood.sample(ns.parameters)
        return loss, metrics, outputs

    def generate(self, n_samples: int = 1, max_timesteps: int = 100, use_mode: bool = False, x=None, u=None, d_0=None, a_0=None,
                 z_0=None, eps=None, uniforms=None):  # fmt: skip
        """Same arguments as the reference (srnn.py:515-535)."""
        x = torch.zeros(n_samples, 1, self.input_size, device=self.device) if x is None else x
        return self.srnn.generate(x=x, u=u, d_0=d_0, a_0=a_0, z_0=z_0, n_samples=n_samples, max_timesteps=max_timesteps,
                                  use_mode=use_mode, eps=eps, uniforms=uniforms)  # fmt: skip
