"""Stochastic Temporal Convolutional Network with the reference's construction API, module tree and outputs
(blvm/models/stcn/stcn.py: `DiagonalGaussianDenseSTCN` :32-76, `STCN` :78-442 — `compute_loss` :247-294, `infer` :296-327,
`forward` :346-431), computed by HIP kernels: the dilated gated residual stack with per-stack skip outputs (K10), the
per-level prior / posterior MLPs (K6), the latent heads with the precision-weighted posterior and the reparameterised
sample (K8b), the per-level KL with free nats (K8), the un-dilated output stack (K10) and the DMoL head (K7).
The model has no recurrence: everything is time-parallel on time-major channel-last tensors [T,B,C].
"""
import math
import warnings
from typing import List, Optional

import torch
import torch.nn as nn

from blvm import ops
from blvm._hip import BlvmHipError
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, KLMetric, LatestMeanMetric, LLMetric, LossMetric
from blvm.models.base_model import BaseModel
from blvm.models.vrnn import LazyNamespace
from blvm.models.wavenet.wavenet_modules import CausalConv1d, ResidualStack
from blvm.modules.convenience import AddConstant
from blvm.modules.distributions import (ConditionalDistribution, DiagonalGaussianDense, DiagonalGaussianMixtureDense,
                                        DiscretizedLogisticMixtureDense)  # fmt: skip


class DiagonalGaussianDenseSTCN(ConditionalDistribution):
    """Two 3-layer MLPs (mean, standard deviation) + softplus_beta + epsilon (stcn.py:32-76)."""

    def __init__(self, in_channels: int, out_channels: int, hidden_channels: int, activation: nn.Module = nn.LeakyReLU,
                 init_sd_mean: float = 1, epsilon: float = 1e-3) -> None:  # fmt: skip
        super().__init__()
        if activation is not nn.LeakyReLU:
            raise NotImplementedError("libblvm_hip: DiagonalGaussianDenseSTCN is built for LeakyReLU")
        self.in_channels, self.out_channels, self.activation = in_channels, out_channels, activation
        self.init_sd_mean, self.epsilon = init_sd_mean, epsilon

        def mlp():
            return nn.Sequential(nn.Linear(in_channels, hidden_channels), activation(), nn.Linear(hidden_channels, hidden_channels),
                                 activation(), nn.Linear(hidden_channels, out_channels))  # fmt: skip

        self.transform_mu = mlp()
        self.transform_sd = mlp()
        self.sd_act = nn.Sequential(nn.Softplus(beta=math.log(2) / (init_sd_mean - epsilon)), AddConstant(epsilon))

    @property
    def softplus_beta(self) -> float:
        return math.log(2) / (self.init_sd_mean - self.epsilon)

    def raw(self, x2d: torch.Tensor):
        """(mu, pre-softplus sd) on [rows, in_channels]; the softplus is applied by the fused latent head (K8b)."""
        out = []
        for seq in (self.transform_mu, self.transform_sd):
            h = ops.mlp(x2d, [seq[0], seq[2]], ops.ACT_LEAKY, seq[1].negative_slope)
            out.append(ops.linear(h, seq[4].weight, seq[4].bias))
        return out


class STCN(BaseModel):
    def __init__(self, likelihood: str = "DMoL", in_channels: int = 1, n_layers: int = 5, n_stacks: Optional[int] = None,
                 latent_size: List[int] = [256, 128, 64, 32, 16], res_channels: int = 256, kernel_size: int = 2,
                 base_dilation: int = 2, n_stack_frames: int = 1, precision_posterior: bool = True, dense: bool = True,
                 top_down: bool = True) -> None:  # fmt: skip
        """Same arguments as the reference (stcn.py:78-123)."""
        super().__init__()
        n_latents = len(latent_size)
        n_stacks = len(latent_size) if n_stacks is None else n_stacks
        if n_stacks != n_latents:
            raise NotImplementedError("libblvm_hip: STCN is built for n_stacks == number of latent variables")
        self.likelihood, self.n_layers, self.n_stacks, self.n_latents = likelihood, n_layers, n_stacks, n_latents
        self.latent_size, self.in_channels, self.res_channels = latent_size, in_channels, res_channels
        self.kernel_size, self.base_dilation, self.n_stack_frames = kernel_size, base_dilation, n_stack_frames
        self.precision_posterior, self.dense, self.top_down = precision_posterior, dense, top_down

        # registration / RNG order of the reference (stcn.py:146-235)
        self.causal = CausalConv1d(in_channels=in_channels * n_stack_frames, out_channels=res_channels, kernel_size=kernel_size)
        self.res_stack = ResidualStack(n_layers=n_layers, n_stacks=n_stacks, res_channels=res_channels, kernel_size=kernel_size,
                                       base_dilation=base_dilation)  # fmt: skip
        self.receptive_fields = [rf + self.causal.kernel_size - 1 for rf in self.res_stack.receptive_fields]
        self.receptive_field = self.receptive_fields[-1]

        prior, posterior = [None] * n_latents, [None] * n_latents
        for i, l in enumerate(reversed(range(n_latents)) if top_down else range(n_latents)):  # (also the creation = RNG order)
            c_in = res_channels if i == 0 else res_channels + latent_size[l + 1 if top_down else l - 1]
            prior[l] = DiagonalGaussianDenseSTCN(c_in, latent_size[l], res_channels, init_sd_mean=0.5)
            posterior[l] = DiagonalGaussianDenseSTCN(c_in, latent_size[l], res_channels, init_sd_mean=0.1)
        self.prior, self.posterior = nn.ModuleList(prior), nn.ModuleList(posterior)

        self.out_transform = ResidualStack(n_layers=n_layers, n_stacks=1, res_channels=res_channels,
                                           in_channels=sum(latent_size) if dense else latent_size[0], kernel_size=kernel_size,
                                           base_dilation=1)  # fmt: skip
        self.inv_std = 1 / math.sqrt(self.n_stacks)

        num_mix = 10
        if likelihood == "DMoL":
            likelihood_module = DiscretizedLogisticMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=num_mix, num_bins=2**16)
        elif likelihood == "GMM":
            likelihood_module = DiagonalGaussianMixtureDense(x_dim=2 * num_mix + num_mix, y_dim=1, num_mix=num_mix, initial_sd=1,
                                                             epsilon=1e-4)  # fmt: skip
        elif likelihood == "Gaussian":
            likelihood_module = DiagonalGaussianDense(x_dim=2, y_dim=1, epsilon=1e-4)
        else:
            raise ValueError(f"Unknown likelihood type {likelihood}")
        self.out_upsample = nn.Sequential(nn.Linear(res_channels, likelihood_module.out_features * n_stack_frames), nn.ReLU())
        self.likelihood_module = likelihood_module

    # ---- inference over the latent hierarchy (stcn.py:296-327) -------------------------------------------------------
    def infer(self, skips, eps, x_sl_dev, B: int, T: int, free_nats: float):
        """skips[l] [T+1,B,C] time-major.  Returns per level (mu_p, sd_p, mu_q, sd_q, z) [T,B,Z_l] and the KL sums."""
        n, S = self.n_latents, self.n_stack_frames
        mu_p, sd_p, mu_q, sd_q, z = ([None] * n for _ in range(5))
        klds, klds_fn = [None] * n, [None] * n
        order = list(reversed(range(n))) if self.top_down else list(range(n))  # bottom-up: each latent conditions on the one below
        for i, l in enumerate(order):
            d_p, d_q = skips[l][:-1], skips[l][1:]  # prior sees frame t-1's features, the posterior frame t's (stcn.py:300-302)
            if i > 0:
                zc = z[l + 1 if self.top_down else l - 1]
                d_p, d_q = torch.cat([d_p, zc], -1), torch.cat([d_q, zc], -1)
            Z = self.latent_size[l]
            mp, sp_raw = self.prior[l].raw(d_p.reshape(T * B, -1))
            mq, sq_raw = self.posterior[l].raw(d_q.reshape(T * B, -1))
            e = eps[l].reshape(T * B, Z)
            sp, mq_c, sq_c, z_l = ops.gauss_latent(mp, sp_raw, mq, sq_raw, e, self.prior[l].softplus_beta,
                                                   self.posterior[l].softplus_beta, self.prior[l].epsilon,
                                                   ops.RSSM_PRECISION if self.precision_posterior else ops.RSSM_PLAIN)  # fmt: skip
            if self.top_down:
                klds[l], klds_fn[l] = ops.gaussian_kl_sums(mq_c, sq_c, mp, sp, x_sl_dev, ops.LAYOUT_TIME_MAJOR, B, T, Z, S, free_nats)
            else:
                # Monte-Carlo KL at the drawn z (stcn.py:286-287, variational.py:73-83): log q(z) - log p(z) per element, masked to the
                # frame stacks that start inside the utterance, free nats shared over the level's Z dimensions; per-utterance sums.
                # (elementwise torch on [T*B, Z]: this mode is not on a BASELINE configuration and has no fused kernel)
                ll = lambda v, mu, sd: -((v - mu) ** 2) / (2 * sd**2) - sd.log() - 0.5 * math.log(2 * math.pi)  # noqa: E731
                kl = (ll(z_l, mq_c, sq_c) - ll(z_l, mp, sp)).view(T, B, Z)
                mask = (torch.arange(T, device=kl.device).unsqueeze(1) * S < x_sl_dev.unsqueeze(0)).unsqueeze(-1)
                kl_fn = torch.clamp(kl, min=free_nats / Z) if free_nats else kl
                klds[l] = (kl * mask).double().sum((0, 2))
                klds_fn[l] = (kl_fn * mask).double().sum((0, 2))
            mu_p[l], sd_p[l], mu_q[l], sd_q[l], z[l] = (t.view(T, B, Z) for t in (mp, sp, mq_c, sq_c, z_l))
        return mu_p, sd_p, mu_q, sd_q, z, klds, klds_fn

    def split_sequence(self, x, x_sl, length: int):
        raise NotImplementedError()

    def forward_split(self, x, x_sl, i_split: int, y=None, eps=None):
        """Receptive-field padding on the first split only (stcn.py:332-342); `eps` as in `forward`."""
        return self.forward(x, x_sl, y=y, pad_receptive_field=(i_split == 0), eps=eps)

    def forward(self, x, x_sl, y=None, pad_receptive_field: bool = True, free_nats: float = 0, beta: float = 1,
                eps: Optional[List[torch.Tensor]] = None):  # fmt: skip
        """x [B,T] in [-1,1]; x_sl [B] (host ints).  `eps[l]` [T',B,z_l] optionally supplies the reparameterisation noise
        (otherwise drawn on the device, top level first as in the reference)."""
        if x.ndim == 3:
            x = x.squeeze(-1)
        if not x.is_cuda:
            raise BlvmHipError("blvm HIP kernels were handed a CPU tensor (no CPU fallback)")
        dev, S, rf, C = x.device, self.n_stack_frames, self.receptive_field, self.res_channels
        lik = self.likelihood_module
        x = x.to(torch.float32)
        x_sl_host = x_sl.detach().cpu().to(torch.int64)
        if y is None:
            y = x.detach()
            if not pad_receptive_field:
                y = y[:, rf * S :]
        y = y.reshape(y.size(0), -1).contiguous()
        B, T_x = x.shape
        Tp = (T_x + S - 1) // S
        xs = torch.nn.functional.pad(x, (0, Tp * S - T_x)) if Tp * S != T_x else x
        xt = xs.view(B, Tp, S).transpose(0, 1).contiguous()  # time-major stacked frames [T',B,S]
        if pad_receptive_field:
            T = Tp
            xt = torch.cat([torch.zeros(rf, B, S, device=dev), xt], 0)
            if T < rf:
                warnings.warn(f"Padded input of {T} frames with a larger receptive_field={rf}.")
        else:
            T = Tp - rf
            x_sl_host = x_sl_host - S * rf
            if Tp <= rf:
                raise ValueError(f"Input must be at least as long as the receptive field if {pad_receptive_field=}")
        T_y = y.size(1)
        mask_len = ops.upload_i32(x_sl_host.clamp(min=0, max=T_y), dev)

        out = self.causal.forward_tm(xt, pad_causal=False)  # [T + rf - 1, B, C]
        n = self.n_latents
        # `d[n_latents - 1 :: n_latents]` (stcn.py:299): every n_latents-th skip connection, the first n_latents of them are
        # used — with n_layers == n_latents (the default) that is the last block of every stack
        n_blocks = len(self.res_stack.dilations)
        if n_blocks // n < n:
            raise IndexError(f"STCN needs n_layers * n_stacks >= n_latents**2 skip connections, got {n_blocks} for {n} latents")
        groups = [(i // n) if (i % n == n - 1 and i // n < n) else -1 for i in range(n_blocks)]
        skips = self.res_stack.forward_tm(out, T + 1, groups=groups)

        if eps is None:
            eps = [None] * n
            for l in (reversed(range(n)) if self.top_down else range(n)):  # the reference's draw order
                eps[l] = torch.randn(T, B, self.latent_size[l], device=dev)
        mu_p, sd_p, mu_q, sd_q, z, klds, klds_fn = self.infer(skips, [e.to(device=dev, dtype=torch.float32).contiguous() for e in eps],
                                                              mask_len, B, T, free_nats)  # fmt: skip

        logits_in = torch.cat(z, -1) if self.dense else z[0]
        ot = self.out_transform
        logits_in = torch.cat([torch.zeros(ot.receptive_field - 1, B, logits_in.size(-1), device=dev), logits_in], 0)
        skip_sum = ot.forward_tm(logits_in, T)  # [T,B,C]: sum of the output blocks' skips
        h = ops.scale_act(skip_sum.view(T * B, C), self.inv_std, 1.0)  # * inv_std (slope 1: no activation)
        up = self.out_upsample[0]
        dec = ops.linear(h, up.weight, up.bias, ops.ACT_RELU)  # [T*B, S*F]
        log_prob = lik.fused_log_prob(dec, y, mask_len, ops.LAYOUT_TIME_MAJOR, B, T_y, T, S)  # K7 / K7b / K7c

        kld, kld_fn = sum(klds), sum(klds_fn)
        n_frames = float(x_sl_host.sum())
        elbo = log_prob - kld
        loss = -(log_prob - beta * kld_fn).sum() / n_frames
        metrics = self.build_metrics(loss, elbo, log_prob, kld, klds, x_sl_host, beta, free_nats)

        F = lik.out_features

        def params():
            d = dec.detach().view(T, B, S, F).permute(1, 0, 2, 3).reshape(B, T * S, F)[:, :T_y]
            return lik(d.contiguous())

        bt = lambda ts: [t.transpose(0, 1) for t in ts]  # noqa: E731  (reference layout [B,T,Z])
        lazy = dict(params=params, reconstructions=lambda ns: lik.sample(ns.params), reconstructions_mode=lambda ns: lik.mode(ns.params))
        output = LazyNamespace(lazy, loss=loss, elbo=elbo, klds=klds, log_prob=log_prob, z=bt(z),
                               z_sl=[torch.ceil(x_sl_host / S).long()] * self.n_stacks, enc_mus=bt(mu_q), prior_mus=bt(mu_p),
                               y=y.unsqueeze(-1))  # fmt: skip
        return loss, metrics, output

    def build_metrics(self, loss, elbo, log_prob, kld, klds, x_sl, beta, free_nats):
        """Metric names / reductions of stcn.py:208-245."""
        n, B = self.n_latents, elbo.numel()
        sums = DeferredScalars(torch.stack([loss.detach().double(), elbo.detach().sum(), log_prob.detach().sum(), kld.detach().sum()]
                                           + [k.detach().sum() for k in klds]))  # fmt: skip
        ln2, nx = math.log(2), float(x_sl.sum())
        nz = float(torch.div(x_sl, self.n_stack_frames, rounding_mode="floor").sum())
        return [
            LossMetric(sums[0], weight_by=B),
            BitsPerDimMetric(sums[1], name="elbo (bpx)", reduce_by=nx),
            LLMetric(sums[1], name="elbo (nats)", reduce_by=B),
            LatestMeanMetric(beta, name="beta"),
            LatestMeanMetric(free_nats, name="free_nats"),
            LLMetric(sums[2], name="rec (nats)", reduce_by=B, log_to_console=False),
            BitsPerDimMetric(sums[2], name="rec (bpx)", reduce_by=nx),
            KLMetric(sums[3], name="kl (nats)", reduce_by=B, log_to_console=False),
            KLMetric(sums[3] / ln2, name="kl (bpz)", reduce_by=nz),
            *[KLMetric(sums[4 + l], name=f"kl_{l} (nats)", reduce_by=B, log_to_console=False) for l in range(n)],
            *[KLMetric(sums[4 + l] / ln2, name=f"kl_{l} (bpz)", reduce_by=nz) for l in range(n)],
            *[KLMetric(sums[4 + l] / ln2, name=f"kl_{l} (bpx)", reduce_by=nx) for l in range(n)],
        ]

    def generate(self, n_samples: int = 1, max_timesteps: int = 100, use_mode_observations: bool = False, x=None):
        raise NotImplementedError()
