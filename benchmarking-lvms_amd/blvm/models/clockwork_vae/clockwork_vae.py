"""Clockwork VAE with the reference's construction API, module tree and outputs (blvm/models/clockwork_vae/
clockwork_vae.py: `CWVAE` :31-393 — `compute_elbo` :132-161, `forward` :200-338 — and `CWVAEAudio` :396-529), computed by
HIP kernels: strided depthwise-separable conv encoder and transposed-conv context decoders (K11 + K6), one RSSM cell
sequence per level incl. BPTT and the per-level KL with scaled free nats (K5 + K8), DMoL head (K7).

Everything between the input and the loss lives time-major channel-last [L,B,C] in HBM; the three nested Python loops of
the reference (levels x steps x per-example state gather) become 3 sequence launches + conv stacks per forward.
"""
import math
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from blvm import ops
from blvm._hip import BlvmHipError
from blvm.evaluation import (BitsPerDimMetric, DeferredScalars, EMAMetric, KLMetric, LatestMeanMetric, LLMetric,
                             LossMetric)  # fmt: skip
from blvm.models.base_model import BaseModel
from blvm.models.clockwork_vae.convolutional_coders import ConvCoder1d
from blvm.models.vrnn import LazyNamespace
from blvm.modules.distributions import DiagonalGaussianDense, DiagonalGaussianMixtureDense, DiscretizedLogisticMixtureDense
from blvm.modules.rssm import RSSMCell
from blvm.utils.operations import split_sequence
from blvm.utils.padding import get_modulo_length, get_modulo_padding, get_same_padding


class CWVAE(nn.Module):
    def __init__(self, z_size: Union[int, List[int]], h_size: Union[int, List[int]], strides: List[int], encoder: nn.Module,
                 decoder: nn.Module, likelihood: nn.Module, g_size: Optional[int] = 0, residual_posterior: bool = False,
                 precision_posterior: bool = False, with_resets: bool = False, jit_compile: bool = True):  # fmt: skip
        """Same arguments as the reference (clockwork_vae.py:32-61).  `jit_compile` is accepted and ignored: the cells run
        as whole-sequence HIP launches, there is nothing to script."""
        super().__init__()
        assert isinstance(strides, list)
        if not (isinstance(encoder, ConvCoder1d) and isinstance(decoder, ConvCoder1d) and decoder.transposed and not encoder.transposed):
            raise NotImplementedError("libblvm_hip: CWVAE needs ConvCoder1d coders (encoder plain, decoder transposed)")
        if not isinstance(likelihood, (DiscretizedLogisticMixtureDense, DiagonalGaussianMixtureDense, DiagonalGaussianDense)):
            raise NotImplementedError("libblvm_hip: CWVAE is built with the DMoL, GMM and Gaussian likelihood heads")

        self.encoder, self.decoder, self.likelihood = encoder, decoder, likelihood
        self.residual_posterior, self.precision_posterior = residual_posterior, precision_posterior
        self.g_size, self.with_resets, self.jit_compile = g_size, with_resets, jit_compile

        self.num_levels = len(strides)
        self.strides = strides
        self.overall_strides = np.cumprod(strides)
        self.overall_stride = self.overall_strides[-1]
        self.overall_receptive_field = self.encoder.overall_receptive_field
        self.overall_receptive_fields = self.encoder.overall_receptive_fields
        self.receptive_fields = self.encoder.receptive_fields

        self.e_size = self.encoder.e_size
        self.z_size = [z_size] * self.num_levels if isinstance(z_size, int) else z_size
        self.h_size = [h_size] * self.num_levels if isinstance(h_size, int) else h_size
        self.c_size = [e_size for e_size in self.decoder.e_size[1:]] + [0]
        assert len(self.z_size) == len(self.h_size) == len(self.c_size), f"{self.z_size=}=={self.h_size=}=={self.c_size=}"

        self.cells = nn.ModuleList(
            RSSMCell(h_dim=h, z_dim=z, c_dim=c, e_dim=e, residual_posterior=residual_posterior, precision_posterior=precision_posterior)
            for h, z, c, e in zip(self.h_size, self.z_size, self.c_size, self.e_size)
        )

    # ---- split evaluation (clockwork_vae.py:163-198) ------------------------------------------------------------------
    def split_sequence(self, x, x_sl, length: int, drop_inactive: bool = False):
        """Splits that are wholly strideable and overlap by rf - stride (what an unsplit conv would have seen)."""
        length = get_modulo_length(length, self.overall_stride, self.overall_receptive_field)
        overlap = self.overall_receptive_field - self.overall_stride
        return split_sequence(x, x_sl, length=length, overlap=overlap, drop_inactive=drop_inactive)

    # ---- with_resets (clockwork_vae.py:273-275, 369-371) -----------------------------------------------------------------
    # Below the top level the state is reset to zeros at every step t with t % strides[l + 1] == 0 (t = 0 included, so a carried
    # state0 is never used there): the level's sequence is T_l / k INDEPENDENT segments of k = strides[l + 1] steps from a zero
    # state.  Independent sequences are what the batch axis of the cell kernels is for: the segments run as (segment, utterance)
    # rows of ONE k-step sequence launch — more rows, fewer dependent steps — and are laid back along time afterwards.
    @staticmethod
    def _fold(t: Optional[torch.Tensor], k: int, nseg: int):
        """[T_l, B, F] -> [k, nseg * B, F] (row = segment * B + utterance), zero-padded to nseg * k steps."""
        if t is None:
            return None
        T_l, B, F = t.shape
        if nseg * k != T_l:
            t = torch.cat([t, t.new_zeros(nseg * k - T_l, B, F)], 0)
        return t.view(nseg, k, B, F).transpose(0, 1).reshape(k, nseg * B, F).contiguous()

    @staticmethod
    def _unfold(t: torch.Tensor, k: int, nseg: int, T_l: int):
        """[k, nseg * B, F] -> [T_l, B, F]."""
        B = t.shape[1] // nseg
        return t.view(k, nseg, B, -1).transpose(0, 1).reshape(nseg * k, B, -1)[:T_l]

    def _resets(self, l: int) -> int:
        return int(self.strides[l + 1]) if (self.with_resets and l < self.num_levels - 1) else 0

    def forward_split(self, x, x_sl, is_last_split: bool, state0=None, beta: float = 1, free_nats: float = 0, y=None,
                      use_mode_global: bool = False, eps=None):  # fmt: skip
        return self.forward(x, x_sl, state0=state0, beta=beta, free_nats=free_nats, y=y, use_mode_global=use_mode_global,
                            pad_strideable=False, pad_same=is_last_split, eps=eps)  # fmt: skip

    # ---- forward ---------------------------------------------------------------------------------------------------------
    def forward(self, x, x_sl, state0: List[Tuple[torch.Tensor, torch.Tensor]] = None, beta: float = 1, free_nats: float = 0,
                y=None, pad_strideable: bool = False, pad_same: bool = True, use_mode_global: bool = False,
                eps: Optional[List[torch.Tensor]] = None):  # fmt: skip
        """x [B,T] or [B,T,1]; x_sl [B] (host ints).  `eps[l]` [T_l,B,z_l] optionally supplies the reparameterisation
        noise of level l (otherwise drawn on the device, top level first as in the reference)."""
        if x.ndim == 3:
            x = x.squeeze(-1)
        if not x.is_cuda:
            raise BlvmHipError("blvm HIP kernels were handed a CPU tensor (no CPU fallback)")
        dev = x.device
        x = x.to(torch.float32)
        NL, os_ = self.num_levels, [int(s) for s in self.overall_strides]

        if pad_strideable and not pad_same:
            x = torch.nn.functional.pad(x, (0, get_modulo_padding(x.shape[1], self.overall_stride, self.overall_receptive_field)))
        y = x.detach() if y is None else (y.squeeze(-1) if y.ndim == 3 else y).to(torch.float32)

        x_sl = x_sl.detach().cpu().to(torch.int64)
        if not pad_same:
            # the reference passes (length, rf, stride) positionally to get_same_padding(length, stride, kernel_size)
            # (clockwork_vae.py:226 vs padding.py:100) — kept.
            consumed = get_same_padding(x.shape[1], self.overall_receptive_field, self.overall_stride)
            x_sl = (x_sl - consumed).clamp(0)
            y = y[:, :-consumed]
        y = y.contiguous()
        B, T = y.shape
        x_sl_dev = ops.upload_i32(x_sl, dev)
        level_sl = [torch.div(x_sl + s - 1, s, rounding_mode="floor") for s in os_]

        same_paddings = []
        for l in range(NL):
            input_length = math.ceil(x.shape[1] / self.strides[l - 1]) if l > 0 else x.shape[1]  # (sic) clockwork_vae.py:245
            same_paddings.append(get_same_padding(input_length, kernel_size=self.receptive_fields[l], stride=self.strides[l]))

        x_tm = x.t().contiguous().unsqueeze(-1)  # [T,B,1]
        encodings = self.encoder.forward_tm(x_tm, pad_right=same_paddings if pad_same else [0] * NL)

        states0 = [None] * NL if state0 is None else state0
        kld_l, kld_fn_l, latents, enc_mus, prior_mus, state_n = ([None] * NL for _ in range(6))
        context = None
        for l in range(NL - 1, -1, -1):
            enc_l = encodings[l]
            T_l = enc_l.shape[0] if (pad_same or context is None) else context.shape[0]
            if enc_l.shape[0] < T_l or (context is not None and context.shape[0] < T_l):
                raise IndexError(f"level {l}: {T_l} steps but only {enc_l.shape[0]} encodings / "
                                 f"{None if context is None else context.shape[0]} context frames")  # fmt: skip
            if int(level_sl[l].max()) > T_l:
                # the reference gathers every example's carried state at step ceil(x_sl / stride_l) - 1 from a list of T_l states
                # (clockwork_vae.py:283-290): beyond it, IndexError.  Without same padding this is EVERY shape: the (sic) positional
                # get_same_padding call above leaves x_sl (nearly) unreduced while the un-padded convolutions shorten the level
                # (tests/golden/split_eval.npz: 98 of 98 probed shapes raise in the reference).  Same error here, before any launch.
                raise IndexError(f"list index out of range (level {l}: state of step {int(level_sl[l].max()) - 1} wanted, the level has {T_l} "
                                 f"steps{'' if pad_same else '; pad_same=False never completes in the reference either'})")  # fmt: skip
            Z = self.z_size[l]
            if eps is not None:
                eps_l = eps[l].to(device=dev, dtype=torch.float32)
            elif use_mode_global:
                eps_l = torch.zeros(T_l, B, Z, device=dev)
            else:
                eps_l = torch.randn(T_l, B, Z, device=dev)
            fn_l = free_nats * os_[l] / os_[0]  # free nats scale with the level's stride (clockwork_vae.py:151)
            k = self._resets(l)
            if k:
                nseg = -(-T_l // k)
                # valid steps of (segment, utterance): the level's step count of the utterance that falls into the segment
                steps = (level_sl[l].unsqueeze(0) - k * torch.arange(nseg).unsqueeze(1)).clamp(0, k)  # [nseg, B]
                seg_sl = ops.upload_i32((steps * os_[l]).reshape(-1), dev)
                f = lambda t: self._fold(t, k, nseg)  # noqa: E731
                u = lambda t: self._unfold(t, k, nseg, T_l)  # noqa: E731
                zs_f, hs_f, kld_f, kld_fn_f, mu_q, _, mu_p, _ = self.cells[l].sequence(
                    f(enc_l[:T_l]), f(None if context is None else context[:T_l]), None, f(eps_l), seg_sl, os_[l], fn_l)  # fmt: skip
                zs = torch.cat([zs_f.new_zeros(1, B, Z), u(zs_f[1:])], 0)
                hs = torch.cat([hs_f.new_zeros(1, B, hs_f.shape[-1]), u(hs_f[1:])], 0)
                kld, kld_fn = kld_f.view(nseg, B).sum(0), kld_fn_f.view(nseg, B).sum(0)
                mu_q, mu_p = u(mu_q), u(mu_p)
            else:
                zs, hs, kld, kld_fn, mu_q, _, mu_p, _ = self.cells[l].sequence(
                    enc_l[:T_l], None if context is None else context[:T_l], states0[l], eps_l, x_sl_dev, os_[l], fn_l)  # fmt: skip
            kld_l[l], kld_fn_l[l] = kld, kld_fn
            latents[l], enc_mus[l], prior_mus[l] = zs[1:].transpose(0, 1), mu_q.transpose(0, 1), mu_p.transpose(0, 1)

            # state to carry into the next split: the one at each example's last valid step (clockwork_vae.py:283-290)
            stop = ops.upload_i32((level_sl[l] - 1).clamp(0) + 1, dev).long()
            rows = torch.arange(B, device=dev)
            state_n[l] = (zs[stop, rows], hs[stop, rows])

            # context for the level below: decode cat(z, h) up to its rate (clockwork_vae.py:292-297)
            _, context = self.decoder.forward_level_tm(torch.cat([zs[1:], hs[1:]], dim=-1), l, pad_right=same_paddings[l])

        if context.shape[0] != T:
            raise BlvmHipError(f"decoded length {context.shape[0]} != target length {T}")
        lik = self.likelihood
        # head: Linear h -> 3*num_mix as a K6 GEMM, then K7 without its fused [F,F] Linear
        par = ops.linear(context.reshape(T * B, -1), lik.params.weight, lik.params.bias)
        log_prob = lik.fused_log_prob(par, y, x_sl_dev, ops.LAYOUT_TIME_MAJOR, B, T, T, 1, fused_linear=False)

        kld, kld_fn = sum(kld_l), sum(kld_fn_l)
        n_frames = float(x_sl.sum())
        elbo = log_prob - kld
        loss = -(log_prob - beta * kld_fn).sum() / n_frames

        metrics = self.build_metrics(loss, elbo, log_prob, kld, kld_l, x_sl, beta, free_nats)

        def parameters():
            return lik(context.detach().transpose(0, 1).contiguous())

        lazy = dict(
            reconstructions_parameters=parameters,
            reconstructions=lambda ns: lik.sample(ns.reconstructions_parameters),
            reconstructions_mode=lambda ns: lik.mode(ns.reconstructions_parameters),
            seq_mask=lambda: torch.arange(int(x_sl.max()), device=dev).unsqueeze(0) < x_sl_dev.unsqueeze(1),
        )
        outputs = LazyNamespace(lazy, elbo=elbo, log_prob=log_prob, kld=kld, y=y.unsqueeze(-1), z=latents, z_sl=level_sl,
                                enc_mus=enc_mus, prior_mus=prior_mus, state_n=state_n)  # fmt: skip
        return loss, metrics, outputs

    def build_metrics(self, loss, elbo, log_prob, kld, kld_l, x_sl, beta, free_nats):
        """Metric names / reductions of clockwork_vae.py:106-130; all sums leave the device in one deferred transfer."""
        NL, B = self.num_levels, elbo.numel()
        sums = DeferredScalars(torch.stack([loss.detach().double(), elbo.detach().sum(), log_prob.detach().sum(), kld.detach().sum()]
                                           + [k.detach().sum() for k in kld_l]))  # fmt: skip
        ln2, n = math.log(2), float(x_sl.sum())
        os_ = [float(s) for s in self.overall_strides]
        return [
            LossMetric(sums[0], weight_by=B),
            EMAMetric(-sums[1] / ln2, name="elbo ema (bpt)", reduce_by=n, weight_by=0.97),
            LLMetric(sums[1], name="elbo (nats)", reduce_by=B),
            BitsPerDimMetric(sums[1], name="elbo (bpt)", reduce_by=n),
            LLMetric(sums[2], name="rec (nats)", reduce_by=B, log_to_console=False),
            BitsPerDimMetric(sums[2], name="rec (bpt)", reduce_by=n),
            KLMetric(sums[3], name="kl (nats)", reduce_by=B, log_to_console=False),
            KLMetric(sums[3] / ln2, name="kl (bpt)", reduce_by=n / os_[0]),
            *[KLMetric(sums[4 + l], name=f"kl_{l} (nats)", reduce_by=B, log_to_console=False) for l in range(NL)],
            *[KLMetric(sums[4 + l] / ln2, name=f"kl_{l} (bpt)", reduce_by=n / os_[l]) for l in range(NL)],
            LatestMeanMetric(beta, name="beta"),
            LatestMeanMetric(free_nats, name="free_nats"),
        ]

    @torch.no_grad()
    def generate(self, n_samples: int = 1, max_timesteps: int = 100, use_mode_observations: bool = False, state0=None,
                 eps: Optional[List[torch.Tensor]] = None):  # fmt: skip
        """Ancestral sampling (clockwork_vae.py:340-393): every level draws z_t from its prior given the decoded context of the
        level above, top-down; the bottom context is decoded to the likelihood parameters.  One K5 launch chain per level
        (posterior branch unused), the K11 context decoders, one head evaluation.  `eps[l]` [T_l,B,z_l] optionally supplies
        the noise.  The reference passes (length, receptive_field, stride) positionally into
        get_same_padding(length, stride, kernel_size) here (:357) — kept, so output lengths match."""
        dev = self.cells[0].prior[0].weight.device
        NL, os_ = self.num_levels, [int(s) for s in self.overall_strides]
        same_paddings = []
        for l in range(NL):
            input_length = math.ceil(max_timesteps / self.strides[l - 1]) if l > 0 else max_timesteps
            same_paddings.append(get_same_padding(input_length, self.receptive_fields[l], self.strides[l]))
        states0 = [None] * NL if state0 is None else state0
        context = None
        for l in range(NL - 1, -1, -1):
            T_l = max_timesteps // os_[l] if l == NL - 1 else context.shape[0]
            if T_l < 1:
                raise IndexError(f"generate: level {l} has no steps for {max_timesteps=}")
            Z = self.z_size[l]
            eps_l = eps[l].to(device=dev, dtype=torch.float32).contiguous() if eps is not None else torch.randn(T_l, n_samples, Z, device=dev)
            k = self._resets(l)
            if k:
                nseg = -(-T_l // k)
                zs_f, hs_f = self.cells[l].generate_sequence(self._fold(context[:T_l], k, nseg), None, self._fold(eps_l, k, nseg), k, nseg * n_samples)
                zs = torch.cat([zs_f.new_zeros(1, n_samples, Z), self._unfold(zs_f[1:], k, nseg, T_l)], 0)
                hs = torch.cat([hs_f.new_zeros(1, n_samples, hs_f.shape[-1]), self._unfold(hs_f[1:], k, nseg, T_l)], 0)
            else:
                zs, hs = self.cells[l].generate_sequence(context, states0[l], eps_l, T_l, n_samples)
            _, context = self.decoder.forward_level_tm(torch.cat([zs[1:], hs[1:]], dim=-1), l, pad_right=same_paddings[l])
        parameters = self.likelihood(context.transpose(0, 1).contiguous())
        x = self.likelihood.mode(parameters) if use_mode_observations else self.likelihood.sample(parameters)
        x_sl = torch.ones(n_samples, dtype=torch.int) * max_timesteps
        return (x, x_sl), SimpleNamespace()


class CWVAEAudio(BaseModel):
    def __init__(self, z_size: Union[int, List[int]] = 64, h_size: Union[int, List[int]] = 128, g_size: Optional[int] = 0,
                 strides: Union[int, List[int]] = [64, 16, 16], dilations: Union[int, List[int]] = 1,
                 residual_posterior: bool = False, precision_posterior: bool = False, num_level_layers: int = 3,
                 stride_per_layer: int = 4, likelihood: str = "dmol", num_mix: int = 10, num_bins: int = 256):  # fmt: skip
        """Same arguments and defaults as the reference (clockwork_vae.py:396-411) — including that only the spelling
        "DMoL" selects the head, so the default "dmol" raises exactly as it does there (:446-447)."""
        super().__init__()
        self.z_size, self.h_size, self.g_size, self.strides, self.dilations = z_size, h_size, g_size, strides, dilations
        self.residual_posterior, self.precision_posterior = residual_posterior, precision_posterior
        self.num_level_layers, self.stride_per_layer = num_level_layers, stride_per_layer
        self.num_mix, self.num_bins = num_mix, num_bins
        self.num_levels = len(strides)

        z_size = [z_size] * self.num_levels if isinstance(z_size, int) else z_size
        h_size = [h_size] * self.num_levels if isinstance(h_size, int) else h_size
        c_size = [h + z + g_size for h, z in zip(h_size, z_size)]
        assert all(h_size[0] == hs for hs in h_size)
        h_size = h_size[0]

        if isinstance(likelihood, str):
            if likelihood == "DMoL":
                likelihood = DiscretizedLogisticMixtureDense(x_dim=h_size, y_dim=1, num_mix=num_mix, num_bins=num_bins)
            elif likelihood == "Gaussian":
                likelihood = DiagonalGaussianDense(x_dim=h_size, y_dim=1, epsilon=1e-2)
            elif likelihood == "GMM":
                likelihood = DiagonalGaussianMixtureDense(x_dim=h_size, y_dim=1, num_mix=num_mix, initial_sd=1, epsilon=1e-2)
            else:
                raise ValueError(f"Unknown likelihood type {likelihood}")
        self.likelihood = likelihood

        encoder = ConvCoder1d(strides=strides, channels_in=1, channels=h_size, kernel_size=5, num_blocks=num_level_layers,
                              stride_per_block=stride_per_layer, transposed=False, block_type="BlockSeparable", activation=nn.ReLU)  # fmt: skip
        decoder = ConvCoder1d(strides=strides, channels_in=c_size, channels=h_size, channels_out=[h_size] + [None] * (self.num_levels - 1),
                              kernel_size=5, num_blocks=num_level_layers, stride_per_block=stride_per_layer, transposed=True,
                              block_type="BlockSeparable", activation=nn.ReLU)  # fmt: skip
        self.cwvae = CWVAE(encoder=encoder, decoder=decoder, likelihood=likelihood, z_size=z_size, h_size=h_size, strides=strides,
                           residual_posterior=residual_posterior, precision_posterior=precision_posterior, g_size=g_size)  # fmt: skip
        self.overall_receptive_field = self.cwvae.overall_receptive_field
        self.overall_stride = self.cwvae.overall_stride
        self.split_sequence = self.cwvae.split_sequence
        self.forward_split = self.cwvae.forward_split

    def forward(self, x, x_sl, state0=None, beta: float = 1, free_nats: float = 0, pad_strideable: bool = True,
                pad_same: bool = True, y=None, eps=None):  # fmt: skip
        return self.cwvae(x, x_sl, state0, beta, free_nats, y, pad_strideable, pad_same, eps=eps)

    def generate(self, *args, **kwargs):
        return self.cwvae.generate(*args, **kwargs)
