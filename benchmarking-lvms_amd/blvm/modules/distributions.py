"""Distribution heads with the reference's constructor signatures and parameter names
(blvm/modules/distributions.py:105-150 DiagonalGaussianDense, :310-387 DiscretizedLogisticMixtureDense).

The modules own the parameters (`params` = nn.Linear, so state_dict keys and initialisation order match the
reference); the arithmetic runs in HIP kernels.  Inside the recurrent models the heads are consumed by the fused
sequence kernels directly and these `forward`s are not on the hot path.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .convenience import AddConstant


class ConditionalDistribution(nn.Module):
    def reset_parameters(self):
        pass


class DiagonalGaussianDense(ConditionalDistribution):
    def __init__(self, x_dim, y_dim, initial_sd: float = 1, epsilon: float = 1e-6):
        super().__init__()
        self.x_dim, self.y_dim, self.initial_sd, self.epsilon = x_dim, y_dim, initial_sd, epsilon
        self.out_features = 2 * y_dim
        self.params = nn.Linear(x_dim, 2 * y_dim)
        # kept for repr/state parity with the reference; the softplus itself is fused into the HIP epilogues
        beta = math.log(2) / (initial_sd - epsilon)
        self.sd_activation = nn.Sequential(nn.Softplus(beta=beta), AddConstant(epsilon)) if epsilon > 0 else nn.Softplus(beta=beta)
        self.reset_parameters()

    @property
    def softplus_beta(self) -> float:
        return math.log(2) / (self.initial_sd - self.epsilon)

    def mode(self, params):
        return params[0].contiguous()

    @torch.no_grad()
    def sample(self, params):
        return params[0] + params[1] * torch.randn_like(params[0])

    def forward(self, x: torch.Tensor):
        lead = x.shape[:-1]
        p = ops.mlp(x.reshape(-1, x.shape[-1]), [self.params], act=ops.ACT_NONE)
        mu, raw = p.view(*lead, -1).chunk(2, dim=-1)
        # softplus on a small [*, y_dim] tensor: torch element-wise (not on the fused hot path)
        return mu, self.sd_activation(raw)

    def fused_log_prob(self, dec, y, x_sl_dev, layout, B, T, Tp, S, fused_linear: bool = True):
        """Masked per-utterance log-likelihood sums [B] straight from decoder activations (K7c): the head's Linear(2->2),
        softplus and `gaussian_ll(..., epsilon=0)` (distributions.py:142-149) in one pass."""
        if self.y_dim != 1:
            raise NotImplementedError("Gaussian likelihood head: y_dim must be 1 on the audio path")
        W, b = (self.params.weight, self.params.bias) if fused_linear else (None, None)
        return ops.gauss_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, self.softplus_beta, self.epsilon)


class DiscretizedLogisticMixtureDense(ConditionalDistribution):
    def __init__(self, x_dim: int, y_dim: int, num_mix: int = 10, num_bins: int = 256, log_epsilon: float = -7.0):
        super().__init__()
        self.x_dim, self.y_dim, self.num_mix, self.num_bins, self.log_epsilon = x_dim, y_dim, num_mix, num_bins, log_epsilon
        self.out_features = num_mix * (2 * y_dim + 1)
        self.params = nn.Linear(x_dim, self.out_features)
        self.reset_parameters()

    def forward(self, x):
        """(logits [*,K], locs [*,1,K], log_scales [*,1,K]) as the reference returns them (distributions.py:381-387)."""
        if self.y_dim != 1:
            raise NotImplementedError("DMoL head: y_dim must be 1 on the audio path")
        lead = x.shape[:-1]
        p = ops.mlp(x.reshape(-1, x.shape[-1]), [self.params], act=ops.ACT_NONE).view(*lead, -1)
        logits = p[..., : self.num_mix]
        locs, log_scales = p[..., self.num_mix :].reshape(*lead, 1, 2 * self.num_mix).chunk(2, dim=-1)
        return logits, locs, log_scales.clamp(min=self.log_epsilon)

    def fused_log_prob(self, dec, y, x_sl_dev, layout, B, T, Tp, S, fused_linear: bool = True):
        """Masked per-utterance log-likelihood sums [B] straight from decoder activations (K7)."""
        W, b = (self.params.weight, self.params.bias) if fused_linear else (None, None)
        return ops.dmol_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, self.num_mix, self.num_bins, self.log_epsilon)

    @staticmethod
    def _pack(params):
        """(logits [*,K], a [*,1,K], b [*,1,K]) -> head-output layout [*, 3K] the kernels read."""
        return torch.cat([params[0], params[1].squeeze(-2), params[2].squeeze(-2)], -1).contiguous()

    @torch.no_grad()
    def mode(self, params):
        return ops.mix_sample(self._pack(params), None, None, 0, self.log_epsilon).unsqueeze(-1)

    @torch.no_grad()
    def sample(self, params, eps: float = 1e-5, uniforms=None):
        """Gumbel-max component pick + logistic sample clamped to [-1,1] (blvm/utils/variational.py:309-349) in one kernel.
        `uniforms` = (u [*,K] in (eps, 1-eps), u2 [*,1] in (1e-8, 1-1e-8)) optionally supplies the two draws the reference
        makes, in its order; otherwise they come from the device RNG."""
        logits = params[0]
        u = torch.empty_like(logits).uniform_(eps, 1.0 - eps) if uniforms is None else uniforms[0].to(logits)
        u2 = torch.empty(*logits.shape[:-1], device=logits.device).uniform_(1e-8, 1.0 - 1e-8) if uniforms is None else uniforms[1].to(logits)
        return ops.mix_sample(self._pack(params), u, u2, 0, self.log_epsilon).unsqueeze(-1)


class DiagonalGaussianMixtureDense(ConditionalDistribution):
    """Gaussian mixture head with the reference's constructor and parameter layout (distributions.py:153-206); note the
    reference's softplus beta here is ln2 / initial_sd (epsilon is NOT subtracted, :168-171)."""

    def __init__(self, x_dim, y_dim, num_mix: int, initial_sd: float = 1, epsilon: float = 1e-6):
        super().__init__()
        self.x_dim, self.y_dim, self.num_mix, self.initial_sd, self.epsilon = x_dim, y_dim, num_mix, initial_sd, epsilon
        self.out_features = num_mix * (2 * y_dim + 1)
        self.params = nn.Linear(x_dim, self.out_features)
        beta = self.softplus_beta
        self.sd_activation = nn.Sequential(nn.Softplus(beta=beta), AddConstant(epsilon)) if epsilon > 0 else nn.Softplus(beta=beta)
        self.reset_parameters()

    @property
    def softplus_beta(self) -> float:
        return math.log(2) / self.initial_sd if self.epsilon > 0 else math.log(2) / (self.initial_sd - self.epsilon)

    def forward(self, x):
        """(logits [*,K], means [*,1,K], sds [*,1,K]) as the reference returns them (distributions.py:190-206)."""
        if self.y_dim != 1:
            raise NotImplementedError("GMM head: y_dim must be 1 on the audio path")
        lead = x.shape[:-1]
        p = ops.mlp(x.reshape(-1, x.shape[-1]), [self.params], act=ops.ACT_NONE).view(*lead, -1)
        logits = p[..., : self.num_mix]
        mu, raw = p[..., self.num_mix :].reshape(*lead, 1, 2 * self.num_mix).chunk(2, dim=-1)
        return logits, mu, self.sd_activation(raw)

    def fused_log_prob(self, dec, y, x_sl_dev, layout, B, T, Tp, S, fused_linear: bool = True):
        """Masked per-utterance log-likelihood sums [B] straight from decoder activations (K7b)."""
        W, b = (self.params.weight, self.params.bias) if fused_linear else (None, None)
        return ops.gmm_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, self.num_mix, self.softplus_beta,
                                self.epsilon if self.epsilon > 0 else 0.0)  # fmt: skip

    @torch.no_grad()
    def mode(self, params):
        return ops.mix_sample(DiscretizedLogisticMixtureDense._pack(params), None, None, 1).unsqueeze(-1)

    @torch.no_grad()
    def sample(self, params, eps: float = 1e-6, noise=None):
        """Gumbel-max component pick + Gaussian sample (blvm/utils/variational.py:156-195) in one kernel; `noise` = (u [*,K]
        uniforms, n [*,1] standard normals) optionally supplies the reference's two draws."""
        logits = params[0]
        u = torch.empty_like(logits).uniform_(eps, 1.0 - eps) if noise is None else noise[0].to(logits)
        nrm = torch.randn(*logits.shape[:-1], device=logits.device) if noise is None else noise[1].to(logits)
        return ops.mix_sample(DiscretizedLogisticMixtureDense._pack(params), u, nrm, 1, sd_beta=0.0).unsqueeze(-1)
