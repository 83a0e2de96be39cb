"""Recurrent state-space cell of the Clockwork-VAE with the reference's constructor and parameter layout
(blvm/modules/rssm.py:18-123); whole sequences run through K5 (`RSSMCell.sequence`)."""
from collections import namedtuple
from typing import Optional, Tuple

import torch
import torch.nn as nn

from blvm import ops
from blvm.modules.distributions import DiagonalGaussianDense

RSSMOutputs = namedtuple("RSSMOutputs", ["z", "enc_mu", "enc_sd", "prior_mu", "prior_sd"])


class RSSMCell(nn.Module):
    def __init__(self, z_dim: int, h_dim: int, c_dim: int, e_dim: int, residual_posterior: bool = False,
                 precision_posterior: bool = False):  # fmt: skip
        super().__init__()
        self.z_dim, self.h_dim, self.e_dim, self.c_dim = z_dim, h_dim, e_dim, c_dim
        self.residual_posterior, self.precision_posterior = residual_posterior, precision_posterior

        def mlp_head(i):
            return nn.Sequential(nn.Linear(i, h_dim), nn.ReLU(), nn.Linear(h_dim, h_dim), nn.ReLU(), nn.Linear(h_dim, h_dim),
                                 nn.ReLU(), DiagonalGaussianDense(h_dim, z_dim))  # fmt: skip

        # registration / RNG order of the reference: gru_in, gru_cell, prior, posterior (rssm.py:45-66)
        self.gru_in = nn.Sequential(nn.Linear(z_dim + c_dim, h_dim), nn.ReLU())
        self.gru_cell = nn.GRUCell(h_dim, h_dim)
        self.prior = mlp_head(h_dim)
        self.posterior = mlp_head(h_dim + e_dim)

    @property
    def mode(self) -> int:
        # residual takes precedence over precision, as in the reference's if/elif (rssm.py:95-98)
        return ops.RSSM_RESIDUAL if self.residual_posterior else (ops.RSSM_PRECISION if self.precision_posterior else ops.RSSM_PLAIN)

    def get_initial_state(self, batch_size: int, device=None):
        device = device if device is not None else self.prior[0].weight.device
        return (torch.zeros(batch_size, self.z_dim, device=device), torch.zeros(batch_size, self.h_dim, device=device))

    def get_empty_context(self, batch_size: int, device=None):
        device = device if device is not None else self.prior[0].weight.device
        return torch.empty(batch_size, 0, device=device)

    def kernel_params(self):
        g = self.gru_cell
        out = [self.gru_in[0].weight, self.gru_in[0].bias, g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh]
        for seq in (self.prior, self.posterior):
            for i in (0, 2, 4):
                out += [seq[i].weight, seq[i].bias]
            out += [seq[6].params.weight, seq[6].params.bias]
        return out

    def sequence(self, enc, ctx, state0, eps, x_sl_dev, stride: int, free_nats: float = 0.0):
        """enc [T,B,E], ctx [T,B,C] or None (C == 0), state0 = (z0, h0) or None.  Returns
        (zs [T+1,B,Z], hs [T+1,B,H], kld, kld_fn, enc_mu, enc_sd, prior_mu, prior_sd)."""
        z0, h0 = state0 if state0 is not None else (None, None)
        return ops.rssm_sequence(enc, ctx, z0, h0, eps, x_sl_dev, self.kernel_params(), self.h_dim, self.z_dim, self.mode, stride,
                                 free_nats, self.prior[6].epsilon)  # fmt: skip

    @torch.no_grad()
    def generate_sequence(self, ctx, state0, eps, T: int, batch_size: int):
        """Ancestral sampling over T steps (`generate`, rssm.py:106-123, stepped by `CWVAE.generate`): z_t ~ prior(h_t).
        ctx [T,B,C] or None, eps [T,B,Z] (zeros = prior mode).  Returns (zs [T+1,B,Z], hs [T+1,B,H])."""
        dev = eps.device
        enc = torch.zeros(T, batch_size, self.e_dim, device=dev)  # the posterior branch is not used by mode 3
        x_sl = torch.full((batch_size,), 2**30, dtype=torch.int32, device=dev)
        z0, h0 = state0 if state0 is not None else (None, None)
        zs, hs, *_ = ops.rssm_sequence(enc, ctx, z0, h0, eps, x_sl, self.kernel_params(), self.h_dim, self.z_dim, ops.RSSM_GENERATE,
                                       1, 0.0, self.prior[6].epsilon)  # fmt: skip
        return zs, hs

    def generate(self, state, context, use_mode: bool = False, eps: Optional[torch.Tensor] = None):
        """Single step of ancestral sampling (rssm.py:106-123)."""
        z, h = state
        B = z.size(0)
        if eps is None:
            eps = torch.zeros(B, self.z_dim, device=z.device) if use_mode else torch.randn(B, self.z_dim, device=z.device)
        ctx = context.unsqueeze(0).contiguous() if context is not None and context.size(-1) > 0 else None
        zs, hs = self.generate_sequence(ctx, (z.contiguous(), h.contiguous()), eps.unsqueeze(0).contiguous(), 1, B)
        e = torch.empty(0, device=z.device)
        return (zs[1], hs[1]), RSSMOutputs(z=zs[1], enc_mu=e, enc_sd=e, prior_mu=None, prior_sd=None)

    def forward(self, enc_inputs: torch.Tensor, state: Tuple[torch.Tensor, torch.Tensor], context: torch.Tensor,
                use_mode: bool = False, eps: Optional[torch.Tensor] = None):  # fmt: skip
        """Single step (rssm.py:79-104) as a length-1 sequence."""
        B = enc_inputs.size(0)
        if eps is None:
            eps = torch.zeros(B, self.z_dim, device=enc_inputs.device) if use_mode else torch.randn(B, self.z_dim, device=enc_inputs.device)
        ctx = context.unsqueeze(0).contiguous() if context is not None and context.size(-1) > 0 else None
        x_sl = torch.ones(B, dtype=torch.int32, device=enc_inputs.device)
        zs, hs, _, _, mq, sq, mp, sp = self.sequence(enc_inputs.unsqueeze(0).contiguous(), ctx, state, eps.unsqueeze(0), x_sl, 1)
        return (zs[1], hs[1]), RSSMOutputs(z=zs[1], enc_mu=mq[0], enc_sd=sq[0], prior_mu=mp[0], prior_sd=sp[0])
