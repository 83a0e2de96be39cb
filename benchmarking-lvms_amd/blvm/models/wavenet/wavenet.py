"""WaveNet with the reference's construction API and state_dict layout (blvm/models/wavenet/wavenet.py:29-293).

forward: left-pad by the receptive field -> causal conv -> in_transform -> 50 gated residual blocks (K10, one autograd
node) -> sum of skips * variance_scale -> ReLU-Linear-ReLU -> likelihood Linear (K6) -> DMoL log-likelihood (K7).
Loss in fp32 like the reference: -sum(ll * mask) / sum(x_sl) (wavenet.py:128-146).
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from blvm import ops
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, LLMetric, LossMetric
from blvm.models.base_model import BaseModel
from blvm.models.vrnn import LazyNamespace
from blvm.modules.distributions import DiscretizedLogisticMixtureDense
from blvm.utils.operations import split_sequence
from blvm.utils.padding import get_modulo_length

from .wavenet_modules import CausalConv1d, PointwiseTransform, ResidualStack


class InputSizeError(Exception):
    def __init__(self, input_size, receptive_field):
        message = "Input size has to be larger than receptive_field\n"
        message += f"Input size: {input_size}, Receptive fields size: {receptive_field}"
        super().__init__(message)


class WaveNet(BaseModel):
    def __init__(self, likelihood: nn.Module, in_channels: int = 1, embedding_dim: int = None, num_bins: int = 256,
                 n_layers: int = 10, n_stacks: int = 5, res_channels: int = 512, skip_channels: Optional[int] = None,
                 gate_channels: Optional[int] = None, kernel_size: int = 2, base_dilation: int = 2, n_stack_frames: int = 1,
                 activation: nn.Module = nn.ReLU):  # fmt: skip
        super().__init__()
        if embedding_dim is not None:
            raise NotImplementedError("libblvm_hip: the embedding input variant of WaveNet is not built (audio runs feed floats)")
        self.n_layers, self.n_stacks, self.in_channels, self.embedding_dim = n_layers, n_stacks, in_channels, embedding_dim
        self.res_channels, self.skip_channels, self.gate_channels = res_channels, skip_channels, gate_channels
        self.kernel_size, self.base_dilation, self.num_bins = kernel_size, base_dilation, num_bins
        self.n_stack_frames, self.activation = n_stack_frames, activation
        self.variance_scale = math.sqrt(1 / self.n_stacks * self.n_layers)
        self.embedding = None
        self.causal = CausalConv1d(in_channels=in_channels * n_stack_frames, out_channels=res_channels, kernel_size=kernel_size)
        self.res_stack = ResidualStack(n_layers=n_layers, n_stacks=n_stacks, res_channels=res_channels, kernel_size=kernel_size,
                                       base_dilation=base_dilation)  # fmt: skip
        self.receptive_field = self.res_stack.receptive_field + self.causal.kernel_size - 1
        self.out_transform = PointwiseTransform(res_channels, res_channels * n_stack_frames)
        self.likelihood = likelihood

    def forward(self, x, x_sl, y=None, pad_causal: bool = True, pad_receptive_field: bool = True):
        lik, nsf, rf = self.likelihood, self.n_stack_frames, self.receptive_field
        if not isinstance(lik, DiscretizedLogisticMixtureDense):
            raise NotImplementedError("libblvm_hip: WaveNet is built with the DMoL likelihood head")
        if x.ndim == 3:
            x = x.squeeze(-1)
        dev = x.device
        x = x.to(torch.float32)
        x_sl_host = x_sl.detach().cpu().to(torch.int64)
        if y is None:
            y = x.detach()
            if not pad_receptive_field:
                y = y[:, rf * nsf :]
        y = y.reshape(y.size(0), -1).contiguous()
        x_sl_strided = (x_sl_host / nsf).ceil().int()
        B = x.size(0)
        if nsf > 1:
            T = x.size(1)
            Tp = (T + nsf - 1) // nsf
            x = torch.nn.functional.pad(x, (0, Tp * nsf - T)).view(B, Tp, nsf)
        else:
            x = x.unsqueeze(-1)
        xt = x.transpose(0, 1).contiguous()  # time-major [T',B,C_in]
        if pad_receptive_field:
            skip_size = xt.size(0)
            xt = torch.cat([torch.zeros(rf, B, xt.size(2), device=dev), xt], 0)
        else:
            skip_size = xt.size(0) - rf
            x_sl_host = x_sl_host - rf
        if xt.size(0) - int(pad_causal) < rf:
            raise InputSizeError(xt.size(0), rf)

        out = self.causal.forward_tm(xt, pad_causal=pad_causal)
        skip_sum = self.res_stack.forward_tm(out, skip_size)  # [skip_size,B,C]
        C = self.res_channels
        logits = self.out_transform.forward_rows(skip_sum.view(skip_size * B, C), self.variance_scale)  # [skip*B, C*nsf]
        par = ops.linear(logits.view(skip_size * B * nsf, C), lik.params.weight, lik.params.bias)  # [skip*B*nsf, 3K]

        T_y = y.size(1)
        mask_len = ops.upload_i32(x_sl_host.clamp(min=0, max=T_y), dev)
        log_prob = ops.dmol_log_prob(par.view(skip_size * B, nsf * lik.out_features), None, None, y, mask_len,
                                     ops.LAYOUT_TIME_MAJOR, B, T_y, skip_size, nsf, lik.num_mix, lik.num_bins,
                                     lik.log_epsilon).to(torch.float32)  # fmt: skip
        n_frames = float(x_sl_host.sum())
        loss = -log_prob.sum() / n_frames

        sums = DeferredScalars(torch.stack([loss.detach().double(), log_prob.detach().double().sum()]))
        metrics = [
            LossMetric(sums[0], weight_by=B),
            LLMetric(sums[1], reduce_by=B),
            BitsPerDimMetric(sums[1], reduce_by=n_frames),
        ]
        F = lik.out_features

        def parameters():
            p = par.detach().view(skip_size, B, nsf, F).permute(1, 0, 2, 3).reshape(B, skip_size * nsf, F)[:, :T_y]
            logits_, rest = p[..., : lik.num_mix], p[..., lik.num_mix :].reshape(B, -1, 1, 2 * lik.num_mix)
            locs, log_scales = rest.chunk(2, dim=-1)
            return logits_, locs, log_scales.clamp(min=lik.log_epsilon)

        def ll_twise():
            ll, _ = ops.dmol_ll_twise(par.detach().view(skip_size * B, nsf * F), None, None, y, mask_len, ops.LAYOUT_TIME_MAJOR, B,
                                      T_y, skip_size, nsf, lik.num_mix, lik.num_bins, lik.log_epsilon)  # fmt: skip
            return ll

        lazy = dict(
            parameters=parameters,
            log_prob_twise=ll_twise,
            predictions=lambda ns: lik.sample(ns.parameters),
            predictions_mode=lambda ns: lik.mode(ns.parameters),
        )
        output = LazyNamespace(lazy, loss=loss, log_prob=log_prob, z=[skip_sum.detach().transpose(0, 1)], z_sl=x_sl_strided,
                               y=y.unsqueeze(-1))  # fmt: skip
        return loss, metrics, output

    def split_sequence(self, x, x_sl, length: int):
        """Overlap = receptive field (wavenet.py:230-242)."""
        overlap = self.receptive_field * self.n_stack_frames
        length = get_modulo_length(length, stride=self.n_stack_frames)
        mode = "extend" if overlap >= length else "consume"
        splits_x, splits_x_sl = split_sequence(x, x_sl, length=length, overlap=overlap, mode=mode)
        if mode == "extend":
            # pad_to_length(split, overlap + length, "left", dim=1) (wavenet.py:240): zeros in front of the TIME axis, whatever follows it
            splits_x = [torch.nn.functional.pad(s, (0, 0) * (s.ndim - 2) + (max(overlap + length - s.size(1), 0), 0)) for s in splits_x]
        return splits_x, splits_x_sl

    def forward_split(self, x, x_sl, i_split: int, y=None):
        return self.forward(x, x_sl, y=y, pad_causal=True, pad_receptive_field=(i_split == 0))

    @torch.no_grad()
    def generate(self, n_samples: int, n_frames: int = 48000, x=None, uniforms=None, cached: bool = False):
        """Sample-by-sample generation from a zero start (wavenet.py:254-293): every frame re-runs the causal conv and the
        whole residual stack over a receptive-field window (no cached sampling, as in the reference), takes the single skip
        output, DIVIDES it by variance_scale (the reference's generate divides where forward multiplies, :274 — kept), applies
        the output transform and the head, samples, and shifts the window (FIFO).  Returns x_hat [B, n_frames, 1].
        `uniforms[t]` optionally supplies the head sampler's two uniform draws of frame t.
        `cached=True` (the reference's TODO, arXiv:1611.09482): the same samples from per-block queues of past activations —
        one new frame per block per step instead of the whole window; with the DMoL head and widths the decode kernel takes
        (K10c: every frame in one launch) through `ops.wavenet_decode`, otherwise block by block (`_generate_cached`)."""
        lik, C, nsf = self.likelihood, self.res_channels, self.n_stack_frames
        if nsf != 1 and (cached or self.in_channels != 1):
            raise NotImplementedError("libblvm_hip: WaveNet.generate on frame stacks is the window path with in_channels=1 (cached=False)")
        if cached:
            if x is not None:
                raise NotImplementedError("libblvm_hip: cached generation starts from the all-zero window")
            if self._decode_kernel_applies():
                return self._generate_decode_kernel(n_samples, n_frames, uniforms)
            return self._generate_cached(n_samples, n_frames, uniforms)
        dev = self.causal.conv.weight.device
        win = torch.zeros(self.receptive_field, n_samples, self.in_channels * nsf, device=dev) if x is None else x.transpose(0, 1).contiguous()
        x_hat = []
        for t in range(n_frames):
            out = self.causal.forward_tm(win, pad_causal=False)
            skip = self.res_stack.forward_tm(out, 1)  # [1,B,C]
            logits = self.out_transform.forward_rows(skip.view(n_samples, C), 1.0 / self.variance_scale)  # [B, C * nsf]
            parameters = lik(logits.view(n_samples, nsf, C))  # unstack_tensor(logits, nsf) (wavenet.py:277-278): one head evaluation per stacked sample
            pred = lik.sample(parameters) if uniforms is None else lik.sample(parameters, uniforms=uniforms[t])  # [B,nsf,1]
            x_hat.append(pred)
            # the nsf new samples are the channels of the next input frame (wavenet.py:290; nsf = 1: one sample, one channel)
            win = torch.cat([win[1:], pred.reshape(1, n_samples, nsf).to(torch.float32)], 0)
        return torch.hstack(x_hat)

    def _decode_kernel_applies(self):
        C, blk, lik = self.res_channels, self.res_stack.res_blocks[0], self.likelihood
        return (isinstance(lik, DiscretizedLogisticMixtureDense) and lik.y_dim == 1 and lik.out_features <= 32 and self.in_channels == 1
                and C % 16 == 0 and blk.skip_channels % 16 == 0 and C <= 128 and blk.skip_channels <= 128 and len(self.res_stack.res_blocks) <= 64)  # fmt: skip

    @torch.no_grad()
    def _generate_decode_kernel(self, n_samples: int, n_frames: int, uniforms=None):
        lik, rs, B = self.likelihood, self.res_stack, n_samples
        dev = self.causal.conv.weight.device
        if uniforms is None:
            u = torch.empty(n_frames, B, lik.num_mix, device=dev).uniform_(1e-5, 1.0 - 1e-5)  # the reference's two draws (variational.py:309-349)
            v = torch.empty(n_frames, B, device=dev).uniform_(1e-8, 1.0 - 1e-8)
        else:
            u = torch.stack([a.reshape(B, lik.num_mix) for a, _ in uniforms[:n_frames]]).to(dev)
            v = torch.stack([b.reshape(B) for _, b in uniforms[:n_frames]]).to(dev)
        t_in, blk = rs.in_transform, rs.res_blocks[0]
        x = ops.wavenet_decode((self.causal.conv.weight, self.causal.conv.bias), (t_in.weight.view(t_in.out_channels, -1), t_in.bias),
                               [b.kernel_params() for b in rs.res_blocks], rs.dilations,
                               (self.out_transform.linear.weight, self.out_transform.linear.bias), (lik.params.weight, lik.params.bias),
                               B, n_frames, blk.inv_std, 1.0 / self.variance_scale, lik.num_mix, lik.log_epsilon, u, v)  # fmt: skip
        return x.unsqueeze(-1)

    @torch.no_grad()
    def _generate_cached(self, n_samples: int, n_frames: int, uniforms=None):
        """Queue-based generation.  A window of zeros is an all-zero past, under which every layer sits at a constant
        activation (its response to zero input, biases included); the queues start from those steady states — one chain of
        single-frame block evaluations with both taps on the same vector — and each step then feeds the last two samples
        through the causal conv and ONE frame through every block (tap 0 = the block's input `dilation` steps ago, from its
        ring buffer; tap 1 = its input now), accumulating the skip branches."""
        lik, C, B = self.likelihood, self.res_channels, n_samples
        dev = self.causal.conv.weight.device
        rs, blocks = self.res_stack, self.res_stack.res_blocks
        S, inv_std = blocks[0].skip_channels, blocks[0].inv_std
        t_in = rs.in_transform

        def causal_pair(x_prev, x_now):  # [B,Cin] each -> causal conv output for the newest position, then the 1x1 in_transform
            c = self.causal.forward_tm(torch.stack([x_prev, x_now], 0), pad_causal=False)  # [1,B,C]
            return ops.linear(c.view(B, C), t_in.weight.view(t_in.out_channels, C), t_in.bias)

        def through_blocks(h, taps0, skip):
            """h [B,C]: input of block 0 now; taps0[i] [B,C]: block i's input `dilation_i` steps ago.  Returns every block's input."""
            inputs = []
            for i, blk in enumerate(blocks):
                inputs.append(h)
                last = i == len(blocks) - 1
                o = ops.wavenet_block_step(torch.stack([taps0[i], h], 0), blk.kernel_params(), inv_std, S, skip, want_output=not last)
                h = o.view(B, C) if o is not None else None
            return inputs

        zero_x = torch.zeros(B, self.in_channels, device=dev)
        # steady state under an all-zero past: block i's delayed input equals its current input
        h, steady = causal_pair(zero_x, zero_x), []
        for i, blk in enumerate(blocks):
            steady.append(h)
            if i < len(blocks) - 1:
                h = ops.wavenet_block_step(torch.stack([h, h], 0), blk.kernel_params(), inv_std, S, torch.zeros(1, B, S, device=dev)).view(B, C)
        queues = [steady[i].unsqueeze(0).repeat(d, 1, 1) for i, d in enumerate(rs.dilations)]  # ring buffers [d_i,B,C]
        heads = [0] * len(blocks)
        x_prev, x_now = zero_x, zero_x
        x_hat = []
        for t in range(n_frames):
            skip = torch.zeros(1, B, S, device=dev)
            taps0 = [queues[i][heads[i]] for i in range(len(blocks))]
            inputs = through_blocks(causal_pair(x_prev, x_now), taps0, skip)
            for i, d in enumerate(rs.dilations):  # overwrite the oldest entry with the current input
                queues[i][heads[i]] = inputs[i]
                heads[i] = (heads[i] + 1) % d
            logits = self.out_transform.forward_rows(skip.view(B, C), 1.0 / self.variance_scale)
            parameters = lik(logits.view(B, 1, C))
            pred = lik.sample(parameters) if uniforms is None else lik.sample(parameters, uniforms=uniforms[t])  # [B,1,1]
            x_hat.append(pred)
            x_prev, x_now = x_now, pred.view(B, 1).to(torch.float32)
        return torch.hstack(x_hat)
