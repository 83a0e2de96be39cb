"""LSTMAudio: deterministic autoregressive baseline with the reference's API (blvm/models/lstm.py:17-141).

stack frames -> embedding MLP (K6) -> 1-layer LSTM with packed-sequence semantics (K4) -> decoder MLP (K6) ->
DMoL next-stack prediction (K7).  Faithful quirks: the loss mask uses x_sl against the SHIFTED target (so a row's
first stack is never scored and up to one stack past its packed length is), and the loss divides by sum(x_sl)
including those unscored frames (lstm.py:111-115).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from blvm import ops
from blvm.evaluation import BitsPerDimMetric, DeferredScalars, LLMetric, LossMetric
from blvm.models.base_model import BaseModel
from blvm.models.vrnn import LazyNamespace
from blvm.modules.distributions import DiscretizedLogisticMixtureDense


class LSTMAudio(BaseModel):
    def __init__(self, stack_size: int = 64, hidden_size: int = 256, num_layers: int = 1, dropout: float = 0,
                 batch_first: bool = True, num_mix: int = 10, num_bins: int = 256):  # fmt: skip
        super().__init__()
        self.stack_size = stack_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.dropout = dropout
        self.batch_first = batch_first
        self.num_mix = num_mix
        self.num_bins = num_bins
        if dropout:
            raise NotImplementedError("libblvm_hip: LSTMAudio is built for dropout=0 (every benchmark run)")

        def mlp(i, o):
            return nn.Sequential(nn.Linear(i, hidden_size), nn.ReLU(), nn.Linear(hidden_size, hidden_size), nn.ReLU(),
                                 nn.Linear(hidden_size, o), nn.ReLU())  # fmt: skip

        self.embedding = mlp(stack_size, hidden_size)
        self.lstm = nn.LSTM(input_size=hidden_size, hidden_size=hidden_size, num_layers=num_layers, bias=True,
                            batch_first=batch_first, dropout=dropout, bidirectional=False, proj_size=0)  # fmt: skip
        self.dropout = None
        self.decoder = mlp(hidden_size, 3 * num_mix * stack_size)
        self.likelihood = DiscretizedLogisticMixtureDense(x_dim=3 * num_mix, y_dim=1, num_mix=num_mix, num_bins=num_bins)

    def forward(self, x: torch.Tensor, x_sl: torch.Tensor, s_0=None):
        S, H, lik = self.stack_size, self.hidden_size, self.likelihood
        dev = x.device
        x_sl_host = x_sl.detach().cpu().to(torch.int64)
        B, T = x.shape
        Tp = (T + S - 1) // S
        L = Tp - 1  # input steps = stacks[:-1], targets = stacks[1:]
        if L < 1:
            raise ValueError("LSTMAudio needs at least two stacks of samples")
        x_sl_stack = (x_sl_host / S).ceil().int()
        xf = x.detach().to(torch.float32)
        xs = torch.nn.functional.pad(xf, (0, Tp * S - T)) if Tp * S != T else xf
        xs = xs.view(B, Tp, S)
        y = xs[:, 1:].reshape(B, L * S).contiguous()  # targets
        inp = xs[:, :-1].transpose(0, 1).contiguous().view(L * B, S)  # time-major inputs
        emb = ops.mlp(inp, [m for m in self.embedding if isinstance(m, nn.Linear)], ops.ACT_RELU, 0.0).view(L, B, H)

        lens = ops.upload_i32((x_sl_stack - 1).clamp(min=0), dev)
        # nn.LSTM(num_layers = n) on packed sequences (lstm.py:93-101): layer l reads layer l-1's outputs (zero beyond a length),
        # every layer with its own carried state; one K4 sequence launch per layer
        out, hns, cns = emb, [], []
        for l in range(self.num_layers):
            h0 = c0 = None
            if s_0 is not None:
                h0, c0 = s_0[0][l].reshape(B, H).contiguous(), s_0[1][l].reshape(B, H).contiguous()
            w = [getattr(self.lstm, f"{n}_l{l}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            out, hn, cn = ops.lstm_sequence(out, h0, c0, lens, *w)
            hns.append(hn)
            cns.append(cn)
        dec = ops.mlp(out.view(L * B, H), [m for m in self.decoder if isinstance(m, nn.Linear)], ops.ACT_RELU, 0.0)

        # mask = arange(L*S) < x_sl  (lstm.py:111): lengths are compared with the SHIFTED target axis
        mask_len = ops.upload_i32(x_sl_host.clamp(max=L * S), dev)
        log_prob = ops.dmol_log_prob(dec, lik.params.weight, lik.params.bias, y, mask_len, ops.LAYOUT_TIME_MAJOR, B, L * S, L,
                                     S, lik.num_mix, lik.num_bins, lik.log_epsilon).to(torch.float32)  # fmt: skip
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
            d = dec.detach().view(L, B, S, F).permute(1, 0, 2, 3).reshape(B, L * S, F)
            return lik(d.contiguous())

        lazy = dict(
            _parameters=parameters,
            reconstruction_sample=lambda ns: lik.sample(ns._parameters),
            reconstruction_mode=lambda ns: lik.mode(ns._parameters),
        )
        outputs = LazyNamespace(lazy, loss=loss, ll=log_prob, z=out.transpose(0, 1), z_sl=x_sl_stack,
                                s_n=(torch.stack(hns), torch.stack(cns)))  # fmt: skip
        return loss, metrics, outputs

    def generate(self, *args, **kwargs):
        raise NotImplementedError()
