"""CPU oracle for the blvm hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32 with fp64 where the reference is fp64) restatement of the arithmetic on the
hot path of JakobHavtorn/benchmarking-lvms (`blvm`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product (`benchmarking-lvms_amd/`) never does.

Parity status: PINNED.  Every function here is checked in `tests/test_oracle_golden.py` against golden
vectors in `tests/golden/` that were produced by importing the unmodified reference in the build container
(`oracle/gen_golden.py`).  The reference's own tests pin only `reverse_sequences` and `CausalConv1d`
(SURVEY.md §4); those known-answer vectors are restated in the tests as well.

All citations are `path:line` relative to the reference checkout.  Everything is functional: models take a
`state_dict` with the reference's key names and explicit noise `eps` instead of drawing from
the global RNG (the reference draws `randn_like` once per step, `blvm/utils/variational.py:141-152`).
"""
import math

import torch
import torch.nn.functional as F

LN2 = math.log(2.0)


# ----------------------------------------------------------------------------------------------------------------------
# data synthesis (blvm/data/transforms.py:192-201 MuLawEncode; SURVEY.md §8d synthetic inputs)
# ----------------------------------------------------------------------------------------------------------------------


def mu_law_encode(u: torch.Tensor, bits: int = 16) -> torch.Tensor:
    """sign(u) * log(1 + mu |u|) / log(mu + 1), mu = 2^bits - 1  (transforms.py:192-201)."""
    mu = 2**bits - 1
    return u.sign() * torch.log(1 + mu * u.abs()) / math.log(mu + 1)


def synth_batch(B: int, T: int, seed: int = 0, ragged: bool = False, bits: int = 16):
    """Synthetic µ-law waveform batch x [B,T] float32 in (-1,1) and lengths x_sl [B] int64."""
    g = torch.Generator().manual_seed(seed)
    u = (torch.rand(B, T, generator=g) * 2 - 1) * 0.5
    x = mu_law_encode(u, bits).to(torch.float32)
    if ragged:
        x_sl = torch.tensor([T - k * (T // (2 * B)) for k in range(B)], dtype=torch.int64)
    else:
        x_sl = torch.full((B,), T, dtype=torch.int64)
    # zero the right padding like the batcher does (batchers.py:120-143 pads with zeros)
    mask = torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1)
    return x * mask, x_sl


# ----------------------------------------------------------------------------------------------------------------------
# shape helpers (blvm/utils/operations.py)
# ----------------------------------------------------------------------------------------------------------------------


def stack_tensor(x: torch.Tensor, stack_size: int):
    """[B,T] -> ([B,ceil(T/s),s], padding): right zero-pad T to a multiple of s (operations.py:14-32, dim=1)."""
    T = x.size(1)
    pad = (-T) % stack_size
    if pad:
        x = F.pad(x, (0, pad))
    return x.reshape(x.size(0), (T + pad) // stack_size, stack_size), pad


def sequence_mask(x_sl: torch.Tensor, max_len: int = None, dtype=torch.bool):
    """arange(T) < x_sl[:,None]  (operations.py:90-119, stride=1)."""
    T = int(max_len) if max_len is not None else int(x_sl.max())
    return (torch.arange(T).unsqueeze(0) < x_sl.unsqueeze(1)).to(dtype)


def reverse_sequences(x: torch.Tensor, x_sl: torch.Tensor):
    """Time-major [T,B,*] per-row reversal that leaves right padding in place (operations.py:56-87)."""
    T = int(x_sl.max())
    out = x.clone()
    for b in range(x.size(1)):
        n = int(x_sl[b])
        out[:n, b] = x[:n, b].flip(0)
    return out[:T] if out.size(0) == T else out


# ----------------------------------------------------------------------------------------------------------------------
# likelihoods (blvm/utils/log_likelihoods.py)
# ----------------------------------------------------------------------------------------------------------------------


def gaussian_ll(y, mu, sd, epsilon: float = 1e-6):
    """Element-wise Gaussian log density (log_likelihoods.py:17-39).

    With epsilon != 0 the reference clamps sd under `torch.no_grad()`, which also DETACHES it: no gradient reaches
    sd on that branch.  The latent heads call with epsilon=0 (distributions.py:140-141)."""
    if epsilon:
        sd = sd.clamp(min=epsilon).detach()
    return -((y - mu) ** 2) / (2 * sd**2) - sd.log() - 0.5 * math.log(2 * math.pi)


def gaussian_mixture_ll(y, logits, mu, sd, epsilon: float = 1e-6):
    """y [*,D], logits [*,K], mu/sd [*,D,K] -> [*,D]; D must be 1 on this path (log_likelihoods.py:42-60)."""
    lp = gaussian_ll(y.unsqueeze(-1), mu, sd, epsilon)  # [*,D,K]
    lp = lp.squeeze(-2) if lp.size(-2) == 1 else lp.sum(-2)  # reduce(D) per component (`reduce`, :10-14)
    return torch.logsumexp(lp + logits.log_softmax(-1), dim=-1)


def dmol_ll(y, logits, locs, log_scales, num_bins: int = 256):
    """Discretized mixture-of-logistics log-likelihood (log_likelihoods.py:170-231).

    y [*,1] in [-1,1]; logits [*,K]; locs, log_scales [*,1,K]  ->  [*].
    Half-bin width 1/(bins-1), edge thresholds 2/bins, fallback offset log(bins/2).
    """
    K = logits.size(-1)
    yk = y.unsqueeze(-1).expand(*y.shape, K)
    c = yk - locs
    inv = torch.exp(-log_scales)
    half = 1.0 / (num_bins - 1)
    plus = inv * (c + half)
    minus = inv * (c - half)
    delta = torch.sigmoid(plus) - torch.sigmoid(minus)
    low = plus - F.softplus(plus)
    high = -F.softplus(minus)
    mid = inv * c
    log_pdf_mid = mid - log_scales - 2.0 * F.softplus(mid)
    inner = torch.where(delta > 1e-5, torch.log(torch.clamp(delta, min=1e-10)), log_pdf_mid - math.log(num_bins / 2))
    lp = torch.where(yk < 2 / num_bins - 1, low, inner)
    lp = torch.where(yk > 1 - 2 / num_bins, high, lp)
    lp = lp.squeeze(-2) if lp.size(-2) == 1 else lp.sum(-2)
    return torch.logsumexp(lp + torch.log_softmax(logits, dim=-1), dim=-1)


# ----------------------------------------------------------------------------------------------------------------------
# variational helpers (blvm/utils/variational.py)
# ----------------------------------------------------------------------------------------------------------------------


def kl_gaussian(mu_q, sd_q, mu_p, sd_p):
    """log sd_p - log sd_q + (sd_q^2 + (mu_q-mu_p)^2) / (2 sd_p^2) - 1/2  (variational.py:67-70)."""
    return sd_p.log() - sd_q.log() + (sd_q.pow(2) + (mu_q - mu_p).pow(2)) / (2 * sd_p.pow(2)) - 0.5


def discount_free_nats(kld, free_nats, shared_last: bool = True):
    """max(kld, free_nats / size(-1)); identity for free_nats in {None, 0}  (variational.py:86-122)."""
    if free_nats is None or free_nats == 0:
        return kld
    floor = free_nats / kld.shape[-1] if shared_last else free_nats
    return torch.maximum(kld, torch.tensor(floor, dtype=kld.dtype))


def precision_weighted_gaussian(mu_1, sd_1, mu_2, sd_2):
    """Product of two Gaussians (variational.py:125-138)."""
    pr_1, pr_2 = sd_1.pow(-2), sd_2.pow(-2)
    var = (pr_1 + pr_2).pow(-1)
    return var * (mu_1 * pr_1 + mu_2 * pr_2), var.sqrt()


# ----------------------------------------------------------------------------------------------------------------------
# heads (blvm/modules/distributions.py)
# ----------------------------------------------------------------------------------------------------------------------


def gaussian_head(x, weight, bias, initial_sd: float = 1.0, epsilon: float = 1e-6):
    """DiagonalGaussianDense.forward: Linear -> chunk -> softplus_beta(s) + eps  (distributions.py:105-150)."""
    p = F.linear(x, weight, bias)
    mu, s = p.chunk(2, dim=-1)
    beta = LN2 / (initial_sd - epsilon)
    sd = F.softplus(s, beta=beta)
    if epsilon > 0:
        sd = sd + epsilon
    return mu, sd


def dmol_head(x, weight, bias, num_mix: int = 10, log_epsilon: float = -7.0):
    """DiscretizedLogisticMixtureDense.forward for y_dim=1 (distributions.py:381-387)."""
    p = F.linear(x, weight, bias)
    logits = p[..., :num_mix]
    rest = p[..., num_mix:].reshape(*p.shape[:-1], 1, 2 * num_mix)
    locs, log_scales = rest.chunk(2, dim=-1)
    return logits, locs, log_scales.clamp(min=log_epsilon)


def dmol_mode(logits, locs):
    """loc of the arg-max-logit component (distributions.py:359-368)."""
    idx = logits.argmax(-1, keepdim=True).unsqueeze(-2)
    return torch.gather(locs, -1, idx).squeeze(-1)


# ----------------------------------------------------------------------------------------------------------------------
# recurrent cells (torch.nn.GRUCell / LSTM semantics, as used at vrnn.py:94,136, srnn.py:113-116, lstm.py:46-55)
# ----------------------------------------------------------------------------------------------------------------------


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """r=s(Wir x+bir+Whr h+bhr), z=s(..), n=tanh(Win x+bin+r*(Whn h+bhn)), h'=(1-z)n+z h; rows [r|z|n]."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """rows [i|f|g|o]; c' = f c + i g; h' = o tanh(c')."""
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, -1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def _mlp(x, sd, prefix, idxs, act):
    for i in idxs:
        x = act(F.linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"]))
    return x


# ----------------------------------------------------------------------------------------------------------------------
# VRNN (blvm/models/vrnn.py)
# ----------------------------------------------------------------------------------------------------------------------


def vrnn_cell_step(sd, x_t, h, eps_t, residual_posterior: bool = True, prefix: str = "vrnn.vrnn_cell"):
    """One VRNNCell.forward (vrnn.py:109-141) with explicit noise."""
    p = _mlp(h, sd, f"{prefix}.prior", (0, 2, 4), F.relu)
    mu_p, sd_p = gaussian_head(p, sd[f"{prefix}.prior.6.params.weight"], sd[f"{prefix}.prior.6.params.bias"])
    q = _mlp(torch.cat([h, x_t], -1), sd, f"{prefix}.posterior", (0, 2, 4), F.relu)
    mu_q, sd_q = gaussian_head(q, sd[f"{prefix}.posterior.6.params.weight"], sd[f"{prefix}.posterior.6.params.bias"])
    if residual_posterior:
        mu_q = mu_q + mu_p
    z = eps_t * sd_q + mu_q
    phi = _mlp(z, sd, f"{prefix}.phi_z", (0, 2, 4, 6), F.relu)
    h_new = gru_cell(
        torch.cat([x_t, phi], -1),
        h,
        sd[f"{prefix}.gru_cell.weight_ih"],
        sd[f"{prefix}.gru_cell.weight_hh"],
        sd[f"{prefix}.gru_cell.bias_ih"],
        sd[f"{prefix}.gru_cell.bias_hh"],
    )
    return h_new, dict(z=z, mu_q=mu_q, sd_q=sd_q, mu_p=mu_p, sd_p=sd_p, phi=phi)


def elbo_terms(ll_twise, kld_twise, x_sl, stride, beta, free_nats, mask_dtype=torch.float64):
    """VRNN.compute_elbo (vrnn.py:255-279): fp64 masks, loss = -sum(log_prob - beta*kld_fn)/sum(x_sl).

    Returns loss, elbo, log_prob, kld (raw), kld_fn (free-nats clamped).
    """
    T = ll_twise.size(1)
    mask = sequence_mask(x_sl, max_len=T, dtype=mask_dtype)
    log_prob = (ll_twise * mask).flatten(1).sum(1)
    mask_kl = mask[:, ::stride].unsqueeze(-1)
    kld = (kld_twise * mask_kl).sum((1, 2))
    elbo = log_prob - kld
    kld_fn = (discount_free_nats(kld_twise, free_nats) * mask_kl).sum((1, 2))
    loss = -(log_prob - beta * kld_fn).sum() / x_sl.sum()
    return loss, elbo, log_prob, kld, kld_fn


def vrnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=0.0, h0=None, stack=64, residual_posterior=True, num_bins=2**16):
    """VRNNAudio(likelihood="DMoL").forward (vrnn.py:281-369, 487-527).  eps: [T',B,z].

    Returns dict(loss, elbo, log_prob, kl, kl_raw, z, h_n, parameters, bpd, metrics...).  `kl` is the
    free-nats-clamped KL, as the reference returns it; `elbo` uses the raw KL.
    """
    B, T = x.shape
    y = x.unsqueeze(-1)
    xs, _ = stack_tensor(x, stack)  # [B,T',s]
    enc = _mlp(xs, sd, "vrnn.encoder", (2, 4, 6), F.leaky_relu)  # [B,T',h]
    Tp = enc.size(1)
    stride = math.ceil(T / Tp)
    r_dim = sd["vrnn.vrnn_cell.gru_cell.weight_hh"].size(1)
    h = torch.zeros(B, r_dim, dtype=x.dtype) if h0 is None else h0
    hs, outs = [h], []
    for t in range(Tp):
        h, o = vrnn_cell_step(sd, enc[:, t], h, eps[t], residual_posterior)
        hs.append(h)
        outs.append(o)
    hs.pop()  # initial state included, last state dropped (vrnn.py:310-311)
    st = lambda k: torch.stack([o[k] for o in outs], 1)  # noqa: E731
    phi, hprev = st("phi"), torch.stack(hs, 1)
    dec = _mlp(torch.cat([phi, hprev], -1), sd, "vrnn.decoder", (0, 2, 4), F.leaky_relu)  # [B,T',s*30]
    dec = dec.reshape(B, Tp * stack, -1)[:, : int(x_sl.max())]
    logits, locs, log_scales = dmol_head(dec, sd["vrnn.likelihood.params.weight"], sd["vrnn.likelihood.params.bias"])
    ll = dmol_ll(y[:, : dec.size(1)], logits, locs, log_scales, num_bins)  # [B,T]
    kld_twise = kl_gaussian(st("mu_q"), st("sd_q"), st("mu_p"), st("sd_p"))  # [B,T',z]
    loss, elbo, log_prob, kld, kld_fn = elbo_terms(ll, kld_twise, x_sl, stride, beta, free_nats)
    return dict(
        loss=loss,
        elbo=elbo,
        log_prob=log_prob,
        kl=kld_fn,
        kl_raw=kld,
        z=st("z"),
        h_n=hs[-1],
        enc=enc,
        ll_twise=ll,
        kld_twise=kld_twise,
        parameters=(logits, locs, log_scales),
        bpd=float((-elbo.detach() / LN2).sum() / x_sl.sum()),
    )


def vrnn_metrics(out, x_sl, beta, free_nats):
    """Metric values as built at vrnn.py:346-355 + metrics.py:209-264 (RunningMeanMetric value = sum/reduce_by)."""
    B = out["elbo"].numel()
    n = float(x_sl.sum())
    return {
        "loss": float(out["loss"]),
        "elbo": float(out["elbo"].sum()) / B,
        "rec": float(out["log_prob"].sum()) / B,
        "kl": float(out["kl"].sum()) / B,
        "kl (bpt)": float((out["kl"] / LN2).sum()) / n,
        "bpd": float((-out["elbo"] / LN2).sum()) / n,
        "beta": float(beta),
        "free_nats": float(free_nats),
    }


# ----------------------------------------------------------------------------------------------------------------------
# annealer (blvm/training/annealers.py:21-72)
# ----------------------------------------------------------------------------------------------------------------------


def cosine_anneal_trace(n_calls, anneal_steps, constant_steps=0, start_value=0.0, end_value=1.0):
    """Values returned by successive CosineAnnealer.step() calls (annealers.py:52-63)."""
    vals = []
    for s in range(1, n_calls + 1):
        if s >= anneal_steps + constant_steps:
            v = end_value
        elif s <= constant_steps:
            v = start_value
        else:
            v = end_value + 0.5 * (start_value - end_value) * (1 + math.cos((s - constant_steps - 1) / anneal_steps * math.pi))
        vals.append(v)
    return vals


# ----------------------------------------------------------------------------------------------------------------------
# LSTMAudio (blvm/models/lstm.py:72-131)
# ----------------------------------------------------------------------------------------------------------------------


def lstm_packed(emb, lens, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None):
    """nn.LSTM on pack_padded_sequence(emb [B,L,H], lens) followed by pad_packed_sequence (lstm.py:96-98): a row past
    its length keeps its state and its outputs are zero.  Returns (out [B,L,H], h_n, c_n)."""
    B, L, _ = emb.shape
    H = w_hh.size(1)
    h = torch.zeros(B, H, dtype=emb.dtype) if h0 is None else h0
    c = torch.zeros(B, H, dtype=emb.dtype) if c0 is None else c0
    outs = []
    for t in range(L):
        live = (t < lens).to(emb.dtype).unsqueeze(-1)
        h2, c2 = lstm_cell(emb[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
        h, c = live * h2 + (1 - live) * h, live * c2 + (1 - live) * c
        outs.append(live * h2)
    return torch.stack(outs, 1), h, c


def lstm_audio_forward(sd, x, x_sl, stack=64, num_mix=10, num_bins=256, s_0=None):
    """LSTMAudio.forward: predict stack t+1 from stacks <= t; loss = -sum(ll * [tau < x_sl]) / sum(x_sl) with the
    mask on the SHIFTED target axis (lstm.py:88-115).  fp32 sums (bool mask), unlike VRNN/SRNN."""
    B, T = x.shape
    xs, _ = stack_tensor(x, stack)
    inp, tgt = xs[:, :-1], xs[:, 1:].reshape(B, -1)
    emb = _mlp(inp, sd, "embedding", (0, 2, 4), F.relu)
    lens = (x_sl / stack).ceil().int() - 1
    h0, c0 = (None, None) if s_0 is None else s_0
    out, h_n, c_n = lstm_packed(emb, lens, sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"], sd["lstm.bias_ih_l0"],
                                sd["lstm.bias_hh_l0"], h0, c0)
    o = _mlp(out, sd, "decoder", (0, 2, 4), F.relu)
    o = o.reshape(B, o.size(1) * stack, 3 * num_mix)
    logits, locs, log_scales = dmol_head(o, sd["likelihood.params.weight"], sd["likelihood.params.bias"], num_mix)
    mask = sequence_mask(x_sl, max_len=tgt.size(1))
    ll = dmol_ll(tgt.unsqueeze(-1), logits, locs, log_scales, num_bins)
    log_prob = (ll * mask).sum(1)
    loss = -log_prob.sum() / x_sl.sum()
    return dict(loss=loss, ll=log_prob, z=out, h_n=h_n, c_n=c_n, bpd=float((-log_prob.detach() / LN2).sum() / x_sl.sum()))


# ----------------------------------------------------------------------------------------------------------------------
# SRNN (blvm/models/srnn.py)
# ----------------------------------------------------------------------------------------------------------------------


def gru_sequence(x, h0, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU over a time-major sequence [T,B,I] (no packing, srnn.py:196): returns (out [T,B,R], h_n)."""
    h, outs = h0, []
    for t in range(x.size(0)):
        h = gru_cell(x[t], h, w_ih, w_hh, b_ih, b_hh)
        outs.append(h)
    return torch.stack(outs, 0), h


def reverse_sequences_diff(x, x_sl):
    """reverse_sequences (operations.py:56-87) as a differentiable gather."""
    T = int(x_sl.max())
    t = torch.arange(T).unsqueeze(1)
    sl = x_sl.unsqueeze(0)
    idx = torch.where(t < sl, sl - 1 - t, t)
    idx = idx.view(T, -1, *([1] * (x.ndim - 2))).expand(-1, -1, *x.shape[2:])
    return torch.gather(x, 0, idx)


def srnn_audio_forward(sd, x, x_sl, eps, beta=1.0, free_nats=0.0, d_0=None, a_0=None, z_0=None, stack=64,
                       residual_posterior=True, smoothing=True, num_bins=2**16):
    """SRNNAudio(likelihood="DMoL").forward (srnn.py:162-302, 456-513) with explicit noise eps [T',B,z].
    `kl` is the RAW KL (srnn.py:156-160); the loss uses the free-nats-clamped one."""
    B, T = x.shape
    y = x.unsqueeze(-1)
    xs, _ = stack_tensor(x, stack)
    enc = _mlp(xs, sd, "srnn.encoder", (2, 4, 6), F.leaky_relu).permute(1, 0, 2)  # [T',B,h]
    Tp = enc.size(0)
    stride = math.ceil(T / Tp)
    x_sl_strided = (x_sl / stride).ceil().long()
    R = sd["srnn.d_forward_recurrent.weight_hh_l0"].size(1)
    Z = sd["srnn.prior.6.params.weight"].size(0) // 2
    u = torch.cat([torch.zeros_like(enc[:1]), enc[:-1]], 0)
    d0 = torch.zeros(B, R, dtype=x.dtype) if d_0 is None else d_0
    g = "srnn.d_forward_recurrent"
    d_seq, d_n = gru_sequence(u, d0, sd[f"{g}.weight_ih_l0"], sd[f"{g}.weight_hh_l0"], sd[f"{g}.bias_ih_l0"], sd[f"{g}.bias_hh_l0"])
    d = torch.cat([d0.unsqueeze(0), d_seq[:-1]], 0)
    cat_xd = torch.cat([enc, d], -1)
    if smoothing:
        a0 = torch.zeros(B, R, dtype=x.dtype) if a_0 is None else a_0
        g = "srnn.a_backward_recurrent"
        a_rev, a_n = gru_sequence(reverse_sequences_diff(cat_xd, x_sl_strided), a0, sd[f"{g}.weight_ih_l0"],
                                  sd[f"{g}.weight_hh_l0"], sd[f"{g}.bias_ih_l0"], sd[f"{g}.bias_hh_l0"])
        a = reverse_sequences_diff(a_rev, x_sl_strided)
    else:
        a, a_n = _mlp(cat_xd, sd, "srnn.a_mlp", (0, 2), F.leaky_relu), None
    z_t = torch.zeros(B, Z, dtype=x.dtype) if z_0 is None else z_0
    zs, mq, sq, mp, sp = [], [], [], [], []
    for t in range(Tp):
        hp = _mlp(torch.cat([d[t], z_t], -1), sd, "srnn.prior", (0, 2, 4), F.leaky_relu)
        mu_p, sd_p = gaussian_head(hp, sd["srnn.prior.6.params.weight"], sd["srnn.prior.6.params.bias"])
        hq = _mlp(torch.cat([a[t], z_t], -1), sd, "srnn.posterior", (0, 2, 4), F.leaky_relu)
        mu_q, sd_q = gaussian_head(hq, sd["srnn.posterior.6.params.weight"], sd["srnn.posterior.6.params.bias"])
        if residual_posterior:
            mu_q = mu_q + mu_p
        z_t = eps[t] * sd_q + mu_q
        zs.append(z_t); mq.append(mu_q); sq.append(sd_q); mp.append(mu_p); sp.append(sd_p)  # noqa: E702
    st = lambda v: torch.stack(v, 1)  # noqa: E731
    z = st(zs)
    dec = _mlp(torch.cat([z, d.permute(1, 0, 2)], -1), sd, "srnn.decoder", (0, 2, 4), F.leaky_relu)
    dec = dec.reshape(B, Tp * stack, -1)[:, : int(x_sl.max())]
    logits, locs, log_scales = dmol_head(dec, sd["srnn.likelihood.params.weight"], sd["srnn.likelihood.params.bias"])
    ll = dmol_ll(y[:, : dec.size(1)], logits, locs, log_scales, num_bins)
    kld_twise = kl_gaussian(st(mq), st(sq), st(mp), st(sp))
    loss, elbo, log_prob, kld, kld_fn = elbo_terms(ll, kld_twise, x_sl, stride, beta, free_nats)
    return dict(loss=loss, elbo=elbo, log_prob=log_prob, kl=kld, kl_fn=kld_fn, z=z, d_n=d_n, a_n=a_n, z_n=zs[-1],
                bpd=float((-elbo.detach() / LN2).sum() / x_sl.sum()))


# ----------------------------------------------------------------------------------------------------------------------
# WaveNet (blvm/models/wavenet/wavenet.py, wavenet_modules.py)
# ----------------------------------------------------------------------------------------------------------------------


def wavenet_dilations(n_layers, n_stacks, base_dilation=2):
    """[1, b, 2b, 4b, ...] per stack (wavenet_modules.py:178-183)."""
    if base_dilation > 1:
        return [1, *[base_dilation * 2**i for i in range(n_layers - 1)]] * n_stacks
    return [1] * n_layers * n_stacks


def wavenet_forward(sd, x, x_sl, n_layers, n_stacks, num_bins=2**16, num_mix=10, base_dilation=2, n_stack_frames=1,
                    pad_causal=True, pad_receptive_field=True):
    """WaveNet.forward for float input (embedding=None) and a DMoL head (wavenet.py:148-228): left-pad by the receptive
    field, causal conv (drop last input, k=2), 1x1 in_transform (always present), gated residual
    blocks without padding (each eats `dilation` frames; skips take the last `skip_size` frames), sum of skips *
    sqrt(n_layers/n_stacks), ReLU-Linear-ReLU, likelihood; fp32 loss = -sum(ll * mask) / sum(x_sl)."""
    dil = wavenet_dilations(n_layers, n_stacks, base_dilation)
    rf = sum(dil) + 1 + (sd["causal.conv.weight"].size(2) - 1)
    y = x.detach()
    if not pad_receptive_field:
        y = y[:, rf * n_stack_frames :]
    if n_stack_frames > 1:
        x, pad = stack_tensor(x, n_stack_frames)
    else:
        x = x.unsqueeze(-1)
    y = y.unsqueeze(-1)
    h = x.transpose(1, 2)  # [B,C,T]
    if pad_receptive_field:
        skip_size = h.size(2)
        h = F.pad(h, (rf, 0))
    else:
        skip_size = h.size(2) - rf
        x_sl = x_sl - rf
    if pad_causal:
        h = h[..., :-1]
    h = F.conv1d(h, sd["causal.conv.weight"], sd["causal.conv.bias"])
    h = F.conv1d(h, sd["res_stack.in_transform.weight"], sd["res_stack.in_transform.bias"])
    C = h.size(1)
    skips = 0
    for i, d in enumerate(dil):
        p = f"res_stack.res_blocks.{i}"
        pre = F.conv1d(h, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], dilation=d)
        a, b = pre.chunk(2, 1)
        rs = F.conv1d(torch.tanh(a) * torch.sigmoid(b), sd[f"{p}.conv1x1rs.weight"], sd[f"{p}.conv1x1rs.bias"])
        r, s = rs[:, :C], rs[:, C:]
        skips = skips + s[..., -skip_size:]
        h = (r + h[..., -r.size(2) :]) * math.sqrt(0.5)
    out = (skips * math.sqrt(n_layers / n_stacks)).transpose(1, 2)
    out = F.relu(F.linear(F.relu(out), sd["out_transform.linear.weight"], sd["out_transform.linear.bias"]))
    if n_stack_frames > 1:
        out = out.reshape(out.size(0), out.size(1) * n_stack_frames, -1)[:, : y.size(1)]
    logits, locs, log_scales = dmol_head(out, sd["likelihood.params.weight"], sd["likelihood.params.bias"], num_mix)
    mask = sequence_mask(x_sl, max_len=y.size(1))
    ll_twise = dmol_ll(y, logits, locs, log_scales, num_bins) * mask
    log_prob = ll_twise.sum(1)
    loss = -log_prob.nansum() / x_sl.nansum()
    return dict(loss=loss, log_prob=log_prob, log_prob_twise=ll_twise, bpd=float((-log_prob.detach() / LN2).sum() / x_sl.sum()))


# ----------------------------------------------------------------------------------------------------------------------
# RSSM cell (blvm/modules/rssm.py:79-104)
# ----------------------------------------------------------------------------------------------------------------------


def rssm_cell_step(sd, enc_t, state, ctx_t, eps_t, residual_posterior=False, precision_posterior=False, prefix=""):
    """One RSSMCell.forward with explicit noise.  state = (z, h); ctx_t [B,C] (C may be 0)."""
    z, h = state
    p = prefix
    g = F.relu(F.linear(torch.cat([z, ctx_t], -1), sd[f"{p}gru_in.0.weight"], sd[f"{p}gru_in.0.bias"]))
    h_new = gru_cell(g, h, sd[f"{p}gru_cell.weight_ih"], sd[f"{p}gru_cell.weight_hh"], sd[f"{p}gru_cell.bias_ih"], sd[f"{p}gru_cell.bias_hh"])
    q = _mlp(torch.cat([h_new, enc_t], -1), sd, f"{p}posterior", (0, 2, 4), F.relu)
    mu_q, sd_q = gaussian_head(q, sd[f"{p}posterior.6.params.weight"], sd[f"{p}posterior.6.params.bias"])
    pr = _mlp(h_new, sd, f"{p}prior", (0, 2, 4), F.relu)
    mu_p, sd_p = gaussian_head(pr, sd[f"{p}prior.6.params.weight"], sd[f"{p}prior.6.params.bias"])
    if residual_posterior:
        mu_q = mu_q + mu_p
    elif precision_posterior:
        mu_q, sd_q = precision_weighted_gaussian(mu_q, sd_q, mu_p, sd_p)
    z_new = eps_t * sd_q + mu_q
    return (z_new, h_new), dict(z=z_new, enc_mu=mu_q, enc_sd=sd_q, prior_mu=mu_p, prior_sd=sd_p)


def rssm_sequence(sd, enc, ctx, state0, eps, **kw):
    """Run the cell over time-major enc [T,B,E] / ctx [T,B,C]; returns stacked (zs, hs [T,B,*]) and distributions."""
    state, zs, hs, ds = state0, [], [], []
    for t in range(enc.size(0)):
        state, d = rssm_cell_step(sd, enc[t], state, ctx[t], eps[t], **kw)
        zs.append(state[0]); hs.append(state[1]); ds.append(d)  # noqa: E702
    st = lambda k: torch.stack([d[k] for d in ds], 0)  # noqa: E731
    return torch.stack(zs, 0), torch.stack(hs, 0), {k: st(k) for k in ("enc_mu", "enc_sd", "prior_mu", "prior_sd")}


# ----------------------------------------------------------------------------------------------------------------------
# Clockwork VAE (blvm/models/clockwork_vae/clockwork_vae.py, convolutional_coders.py)
# ----------------------------------------------------------------------------------------------------------------------


def get_same_padding(length, stride, kernel_size, dilation=1):
    """blvm/utils/padding.py:100-117."""
    return max(0, dilation * (kernel_size - 1) - (length - 1) % stride)


def coder_block_strides(strides, num_blocks, stride_per_block):
    """Stride of every block per level: `stride_per_block` until the level's stride is used up, then 1
    (convolutional_coders.py:178-191)."""
    out = []
    for s in strides:
        rem, lvl = s, []
        for _ in range(num_blocks):
            if rem >= stride_per_block:
                lvl.append(stride_per_block)
                rem //= stride_per_block
            else:
                assert rem == 1
                lvl.append(1)
        out.append(lvl)
    return out


def coder_receptive_fields(block_strides, kernel_size=5):
    """Per-level receptive field r += (k-1)*s_in, s_in *= s (blvm/utils/convolutions.py:117-119)."""
    out = []
    for lvl in block_strides:
        s_in, r = 1, 1
        for s in lvl:
            r += (kernel_size - 1) * s_in
            s_in *= s
        out.append(r)
    return out


def separable_block(sd, p, x, stride, transposed):
    """BlockSeparable (convolutional_coders.py:29-66) on [B,C,T]: 1x1 conv -> ReLU -> GroupNorm(groups=C) -> depthwise
    (transposed) conv -> ReLU -> GroupNorm -> 1x1 conv (no bias), + input (nearest-resampled when the length changed,
    :15-26)."""
    m = f"{p}.block.module"
    h = F.relu(F.conv1d(x, sd[f"{m}.0.weight"], sd[f"{m}.0.bias"]))
    C = h.size(1)
    h = F.group_norm(h, C, sd[f"{m}.2.weight"], sd[f"{m}.2.bias"])
    conv = F.conv_transpose1d if transposed else F.conv1d
    h = F.relu(conv(h, sd[f"{m}.3.depthwise_conv.weight"], sd[f"{m}.3.depthwise_conv.bias"], stride=stride, groups=C))
    h = F.group_norm(h, C, sd[f"{m}.3.norm.weight"], sd[f"{m}.3.norm.bias"])
    h = F.conv1d(h, sd[f"{m}.3.pointwise_conv.weight"])
    if h.size(-1) == x.size(-1):
        return h + x
    return h + F.interpolate(x, size=h.size(-1), mode="nearest")


def coder_level(sd, p, hidden, level, block_strides, transposed, pad_right=0):
    """ConvCoder1d.forward_level (convolutional_coders.py:277-291): in-projection, right zero padding BEFORE the blocks
    (encoder) or cropping AFTER them (transposed decoder), out-projection.  Transposed coders hold their blocks in
    mirrored order (:227-231)."""
    if f"{p}.in_projs.{level}.0.weight" in sd:
        hidden = F.relu(F.conv1d(hidden, sd[f"{p}.in_projs.{level}.0.weight"], sd[f"{p}.in_projs.{level}.0.bias"]))
    if not transposed and pad_right:
        hidden = F.pad(hidden, [0, pad_right])
    strides = block_strides[level][::-1] if transposed else block_strides[level]
    for b, s in enumerate(strides):
        hidden = separable_block(sd, f"{p}.levels.{level}.{b}", hidden, s, transposed)
    if transposed and pad_right:
        hidden = F.pad(hidden, [0, -pad_right])
    enc = hidden
    if f"{p}.out_projs.{level}.0.weight" in sd:
        enc = F.relu(F.conv1d(hidden, sd[f"{p}.out_projs.{level}.0.weight"], sd[f"{p}.out_projs.{level}.0.bias"]))
    return hidden, enc


def cwvae_audio_forward(sd, x, x_sl, eps, strides, num_level_layers, stride_per_layer, beta=1.0, free_nats=0.0, state0=None,
                        residual_posterior=False, precision_posterior=False, num_mix=10, num_bins=256, prefix="cwvae", with_resets=False):
    """CWVAE.forward with pad_same=True (clockwork_vae.py:200-338) for CWVAEAudio (:396-529).  x [B,T] float, x_sl [B];
    eps[l] [T_l,B,z_l].  Levels run top-down; level l's context is the decoded cat(z, h) of level l+1; the KL of level l
    is masked by ceil(x_sl / overall_stride_l) with free nats scaled by overall_stride_l / overall_stride_0 (:147-153);
    reductions in fp32 as the reference (bool masks)."""
    NL = len(strides)
    os_ = [int(v) for v in torch.tensor(strides).cumprod(0)]
    bs = coder_block_strides(strides, num_level_layers, stride_per_layer)
    rfs = coder_receptive_fields(bs)
    B, T = x.shape
    y = x.detach().unsqueeze(-1)
    same = []
    for l in range(NL):
        length = math.ceil(T / strides[l - 1]) if l > 0 else T  # (sic) clockwork_vae.py:245
        same.append(get_same_padding(length, kernel_size=rfs[l], stride=strides[l]))

    hidden, encs = x.unsqueeze(1), []
    for l in range(NL):
        hidden, e = coder_level(sd, f"{prefix}.encoder", hidden, l, bs, False, same[l])
        encs.append(e)

    ctx = None
    kld_l, kld_fn_l, zs_l, hs_l, mus, state_n = [None] * NL, [None] * NL, [None] * NL, [None] * NL, [None] * NL, [None] * NL
    for l in range(NL - 1, -1, -1):
        enc = encs[l].permute(2, 0, 1)  # [T_l,B,E]
        T_l = enc.size(0)
        c = torch.zeros(T_l, B, 0, dtype=x.dtype) if ctx is None else ctx.permute(2, 0, 1)
        cell_sd = {k[len(f"{prefix}.cells.{l}."):]: v for k, v in sd.items() if k.startswith(f"{prefix}.cells.{l}.")}
        Z, H = cell_sd["prior.6.params.weight"].size(0) // 2, cell_sd["gru_cell.weight_hh"].size(1)
        st0 = (torch.zeros(B, Z, dtype=x.dtype), torch.zeros(B, H, dtype=x.dtype)) if state0 is None else state0[l]
        k_reset = int(strides[l + 1]) if (with_resets and l < NL - 1) else 0
        if k_reset:  # clockwork_vae.py:273-275: a zero state whenever the level above ticks (t % strides[l + 1] == 0, t = 0 included)
            parts = []
            for t0 in range(0, T_l, k_reset):
                zero = (torch.zeros(B, Z, dtype=x.dtype), torch.zeros(B, H, dtype=x.dtype))
                parts.append(rssm_sequence(cell_sd, enc[t0:t0 + k_reset], c[t0:t0 + k_reset], zero, eps[l][t0:t0 + k_reset],
                                           residual_posterior=residual_posterior, precision_posterior=precision_posterior))
            zs, hs = torch.cat([q[0] for q in parts], 0), torch.cat([q[1] for q in parts], 0)
            d = {key: torch.cat([q[2][key] for q in parts], 0) for key in parts[0][2]}
        else:
            zs, hs, d = rssm_sequence(cell_sd, enc, c, st0, eps[l], residual_posterior=residual_posterior,
                                      precision_posterior=precision_posterior)
        kl = kl_gaussian(d["enc_mu"], d["enc_sd"], d["prior_mu"], d["prior_sd"])  # [T_l,B,Z]
        sl = torch.ceil(x_sl / os_[l]).to(torch.int64)
        mask = sequence_mask(sl, max_len=T_l).t().unsqueeze(-1)  # [T_l,B,1]
        fn = free_nats * os_[l] / os_[0]
        kld_l[l] = (kl * mask).sum((0, 2))
        kld_fn_l[l] = (discount_free_nats(kl, fn) * mask).sum((0, 2))
        stop = (sl - 1).clamp(0)
        state_n[l] = (torch.stack([zs[t, b] for b, t in enumerate(stop)]), torch.stack([hs[t, b] for b, t in enumerate(stop)]))
        zs_l[l], hs_l[l], mus[l] = zs, hs, (d["enc_mu"], d["prior_mu"])
        _, ctx = coder_level(sd, f"{prefix}.decoder", torch.cat([zs, hs], -1).permute(1, 2, 0), l, bs, True, same[l])

    dec = ctx.permute(0, 2, 1)  # [B,T,h]
    # the head is registered on CWVAEAudio first (`likelihood.*`) and again inside CWVAE (`cwvae.likelihood.*`, same tensor)
    logits, locs, log_scales = dmol_head(dec, sd["likelihood.params.weight"], sd["likelihood.params.bias"], num_mix)
    seq_mask = sequence_mask(x_sl, max_len=T)
    ll_twise = dmol_ll(y, logits, locs, log_scales, num_bins) * seq_mask
    log_prob = ll_twise.view(B, -1).sum(1)
    kld, kld_fn = sum(kld_l), sum(kld_fn_l)
    elbo = log_prob - kld
    loss = -(log_prob - beta * kld_fn).sum() / x_sl.sum()
    return dict(loss=loss, elbo=elbo, log_prob=log_prob, kld=kld, kld_l=kld_l, z=zs_l, h=hs_l, mus=mus, state_n=state_n,
                dec=dec, encodings=encs, bpd=float((-elbo.detach() / LN2).sum() / x_sl.sum()))


# ----------------------------------------------------------------------------------------------------------------------
# STCN (blvm/models/stcn/stcn.py)
# ----------------------------------------------------------------------------------------------------------------------


def residual_stack_skips(sd, p, h, dilations, skip_size):
    """ResidualStack.forward (wavenet_modules.py:195-215): 1x1 in_transform, then gated residual blocks without padding;
    returns the LIST of skip tensors, each cut to its last `skip_size` frames.  h [B,C,T]."""
    h = F.conv1d(h, sd[f"{p}.in_transform.weight"], sd[f"{p}.in_transform.bias"])
    C = sd[f"{p}.res_blocks.0.conv.weight"].size(1)
    skips = []
    for i, d in enumerate(dilations):
        b = f"{p}.res_blocks.{i}"
        pre = F.conv1d(h, sd[f"{b}.conv.weight"], sd[f"{b}.conv.bias"], dilation=d)
        a, g = pre.chunk(2, 1)
        rs = F.conv1d(torch.tanh(a) * torch.sigmoid(g), sd[f"{b}.conv1x1rs.weight"], sd[f"{b}.conv1x1rs.bias"])
        r, s = rs[:, :C], rs[:, C:]
        skips.append(s[..., -skip_size:])
        h = (r + h[..., -r.size(2):]) * math.sqrt(0.5)
    return skips


def stcn_gaussian(sd, p, x, init_sd_mean, epsilon=1e-3):
    """DiagonalGaussianDenseSTCN.forward (stcn.py:32-76): two 3-layer LeakyReLU MLPs, sd = softplus_beta(.) + epsilon."""
    def mlp(q):
        h = F.leaky_relu(F.linear(x, sd[f"{q}.0.weight"], sd[f"{q}.0.bias"]))
        h = F.leaky_relu(F.linear(h, sd[f"{q}.2.weight"], sd[f"{q}.2.bias"]))
        return F.linear(h, sd[f"{q}.4.weight"], sd[f"{q}.4.bias"])
    beta = math.log(2) / (init_sd_mean - epsilon)
    return mlp(f"{p}.transform_mu"), F.softplus(mlp(f"{p}.transform_sd"), beta=beta) + epsilon


def stcn_forward(sd, x, x_sl, eps, n_layers, latent_size, n_stack_frames=1, base_dilation=2, beta=1.0, free_nats=0.0,
                 precision_posterior=True, dense=True, num_mix=10, num_bins=2**16, top_down=True, pad_receptive_field=True):
    """STCN.forward, DMoL head (stcn.py:346-431).  pad_receptive_field=False (`forward_split(i_split > 0)`, stcn.py:332-342, 366-369,
    383-387): no left padding, the first rf * S samples are conditioned on only (dropped from y), T' - rf latent steps, and x_sl is
    reduced by rf * S WITHOUT a clamp — a negative length masks everything and still enters the loss's normaliser.  x [B,T]; eps[l] [B,T',z_l] (one draw per level in
    the order the levels are visited, stcn.py:325).  top_down=False (stcn.py:165-170, 284-287, 310-316): the levels are visited
    bottom first, each conditions on the latent BELOW it, and the KL is the Monte-Carlo estimate log q(z) - log p(z).
    fp32 reductions with bool masks as the reference."""
    n = len(latent_size)
    S = n_stack_frames
    dil = wavenet_dilations(n_layers, n, base_dilation)
    rf = sum(dil) + 1 + (sd["causal.conv.weight"].size(2) - 1)
    y = x.detach()
    if not pad_receptive_field:
        y = y[:, rf * S :]
    y = y.unsqueeze(-1)
    if S > 1:
        xs, pad = stack_tensor(x, S)
    else:
        xs = x.unsqueeze(-1)
    h = xs.transpose(1, 2)  # [B,C,T']
    if pad_receptive_field:
        T = h.size(2)
        h = F.pad(h, (rf, 0))
    else:
        T = h.size(2) - rf
        x_sl = x_sl - S * rf
        if h.size(2) <= rf:
            raise ValueError("Input must be at least as long as the receptive field if pad_receptive_field=False")
    h = F.conv1d(h, sd["causal.conv.weight"], sd["causal.conv.bias"])
    d = residual_stack_skips(sd, "res_stack", h, dil, T + 1)[n - 1 :: n]  # the last skip of every stack (stcn.py:299)
    d_p = [t[..., :-1].permute(0, 2, 1) for t in d]
    d_q = [t[..., 1:].permute(0, 2, 1) for t in d]
    mu_p, sd_p, mu_q, sd_q, z = ([None] * n for _ in range(5))
    for i, l in enumerate(reversed(range(n)) if top_down else range(n)):
        lc = l + 1 if top_down else l - 1
        in_p, in_q = (d_p[l], d_q[l]) if i == 0 else (torch.cat([d_p[l], z[lc]], -1), torch.cat([d_q[l], z[lc]], -1))
        mu_p[l], sd_p[l] = stcn_gaussian(sd, f"prior.{l}", in_p, 0.5)
        mu_q[l], sd_q[l] = stcn_gaussian(sd, f"posterior.{l}", in_q, 0.1)
        if precision_posterior:
            mu_q[l], sd_q[l] = precision_weighted_gaussian(mu_p[l], sd_p[l], mu_q[l], sd_q[l])
        z[l] = eps[l] * sd_q[l] + mu_q[l]
    logits_in = (torch.cat(z, -1) if dense else z[0]).permute(0, 2, 1)
    out_dil = [1] * n_layers
    logits_in = F.pad(logits_in, (sum(out_dil), 0))  # out_transform.receptive_field - 1
    logits = sum(residual_stack_skips(sd, "out_transform", logits_in, out_dil, T)) * (1 / math.sqrt(n))
    logits = F.relu(F.linear(logits.permute(0, 2, 1), sd["out_upsample.0.weight"], sd["out_upsample.0.bias"]))
    if S > 1:
        logits = logits.reshape(logits.size(0), logits.size(1) * S, -1)[:, : y.size(1)]
    lg, locs, log_scales = dmol_head(logits, sd["likelihood_module.params.weight"], sd["likelihood_module.params.bias"], num_mix)
    seq_mask = sequence_mask(x_sl, max_len=y.size(1))
    log_prob = (dmol_ll(y, lg, locs, log_scales, num_bins) * seq_mask).sum(1)
    z_mask = seq_mask[:, ::S].unsqueeze(-1)
    if top_down:
        kl = [kl_gaussian(mu_q[l], sd_q[l], mu_p[l], sd_p[l]) * z_mask for l in range(n)]
    else:  # kl_divergence_gaussian_mc (variational.py:73-83)
        kl = [(gaussian_ll(z[l], mu_q[l], sd_q[l], 0) - gaussian_ll(z[l], mu_p[l], sd_p[l], 0)) * z_mask for l in range(n)]
    kl_fn = [discount_free_nats(kl[l], free_nats) * z_mask for l in range(n)]
    kld, kld_fn = torch.cat(kl, -1).sum((1, 2)), torch.cat(kl_fn, -1).sum((1, 2))
    elbo = log_prob - kld
    loss = -(log_prob - beta * kld_fn).sum() / x_sl.sum()
    return dict(loss=loss, elbo=elbo, log_prob=log_prob, kld=kld, klds=[k.sum((1, 2)) for k in kl], z=z, mu_q=mu_q, mu_p=mu_p,
                bpd=float((-elbo.detach() / LN2).sum() / x_sl.sum()))


def rssm_generate_step(sd, state, ctx_t, eps_t, prefix=""):
    """RSSMCell.generate (rssm.py:106-123): GRU update from cat[z, context], z ~ prior(h)."""
    z, h = state
    p = prefix
    g = F.relu(F.linear(torch.cat([z, ctx_t], -1), sd[f"{p}gru_in.0.weight"], sd[f"{p}gru_in.0.bias"]))
    h_new = gru_cell(g, h, sd[f"{p}gru_cell.weight_ih"], sd[f"{p}gru_cell.weight_hh"], sd[f"{p}gru_cell.bias_ih"], sd[f"{p}gru_cell.bias_hh"])
    pr = _mlp(h_new, sd, f"{p}prior", (0, 2, 4), F.relu)
    mu_p, sd_p = gaussian_head(pr, sd[f"{p}prior.6.params.weight"], sd[f"{p}prior.6.params.bias"])
    return (eps_t * sd_p + mu_p, h_new)


def cwvae_audio_generate(sd, eps, n_samples, max_timesteps, strides, num_level_layers, stride_per_layer, num_mix=10, prefix="cwvae"):
    """CWVAE.generate (clockwork_vae.py:340-393) up to the likelihood parameters; eps[l] [T_l,B,z_l] (zeros = prior means).
    The same-padding call passes (length, receptive_field, stride) positionally into (length, stride, kernel_size) (:357)."""
    NL = len(strides)
    os_ = [int(v) for v in torch.tensor(strides).cumprod(0)]
    bs = coder_block_strides(strides, num_level_layers, stride_per_layer)
    rfs = coder_receptive_fields(bs)
    same = []
    for l in range(NL):
        length = math.ceil(max_timesteps / strides[l - 1]) if l > 0 else max_timesteps
        same.append(get_same_padding(length, rfs[l], strides[l]))  # (sic) stride=rf, kernel_size=stride
    ctx, zs_l = None, [None] * NL
    for l in range(NL - 1, -1, -1):
        cell_sd = {k[len(f"{prefix}.cells.{l}."):]: v for k, v in sd.items() if k.startswith(f"{prefix}.cells.{l}.")}
        Z, H = cell_sd["prior.6.params.weight"].size(0) // 2, cell_sd["gru_cell.weight_hh"].size(1)
        T_l = max_timesteps // os_[l] if ctx is None else ctx.size(2)
        c = torch.zeros(T_l, n_samples, 0) if ctx is None else ctx.permute(2, 0, 1)
        state, zs, hs = (torch.zeros(n_samples, Z), torch.zeros(n_samples, H)), [], []
        for t in range(T_l):
            state = rssm_generate_step(cell_sd, state, c[t], eps[l][t])
            zs.append(state[0]); hs.append(state[1])  # noqa: E702
        zs_l[l] = torch.stack(zs, 0)
        _, ctx = coder_level(sd, f"{prefix}.decoder", torch.cat([zs_l[l], torch.stack(hs, 0)], -1).permute(1, 2, 0), l, bs, True, same[l])
    dec = ctx.permute(0, 2, 1)
    logits, locs, log_scales = dmol_head(dec, sd["likelihood.params.weight"], sd["likelihood.params.bias"], num_mix)
    return dict(logits=logits, locs=locs, log_scales=log_scales, mode=dmol_mode(logits, locs), z=zs_l)


def vrnn_audio_generate(sd, n_samples, max_timesteps, stack, eps, num_mix=10, prefix="vrnn"):
    """VRNNAudio.generate with use_mode semantics for the observations (vrnn.py:371-434, 529-546): frame stack x_t -> encoder ->
    `VRNNCell.generate` (z = mu_p + sd_p * eps_t; :143-164) -> decoder(cat[phi_z, h_NEW]) -> DMoL head -> mode, fed back.
    eps [T,B,z] (zeros = the reference's use_mode=True).  Returns x [B, 1+T, stack] (first frame = the zero start frame)."""
    cell = f"{prefix}.vrnn_cell"
    H = sd[f"{cell}.prior.0.weight"].size(0)
    R = sd[f"{cell}.gru_cell.weight_hh"].size(1)
    x = torch.zeros(n_samples, stack)
    h = torch.zeros(n_samples, R)
    frames = [x]
    for t in range(max_timesteps):
        enc = _mlp(x, sd, f"{prefix}.encoder", (2, 4, 6), F.leaky_relu)
        p = _mlp(h, sd, f"{cell}.prior", (0, 2, 4), F.relu)
        mu_p, sd_p = gaussian_head(p, sd[f"{cell}.prior.6.params.weight"], sd[f"{cell}.prior.6.params.bias"])
        z = eps[t] * sd_p + mu_p
        phi = _mlp(z, sd, f"{cell}.phi_z", (0, 2, 4, 6), F.relu)
        h = gru_cell(torch.cat([enc, phi], -1), h, sd[f"{cell}.gru_cell.weight_ih"], sd[f"{cell}.gru_cell.weight_hh"],
                     sd[f"{cell}.gru_cell.bias_ih"], sd[f"{cell}.gru_cell.bias_hh"])
        dec = _mlp(torch.cat([phi, h], -1), sd, f"{prefix}.decoder", (0, 2, 4), F.leaky_relu).view(n_samples, stack, 3 * num_mix)
        logits, locs, _ = dmol_head(dec, sd[f"{prefix}.likelihood.params.weight"], sd[f"{prefix}.likelihood.params.bias"], num_mix)
        x = dmol_mode(logits, locs).squeeze(-1)
        frames.append(x)
    return torch.stack(frames, 1)


def dmol_sample(logits, locs, log_scales, u, u2):
    """rsample_discretized_logistic_mixture (variational.py:309-349) with its two uniform draws made explicit:
    u [*,K] in (1e-5, 1-1e-5) -> Gumbel-max component pick; u2 [*,1] in (1e-8, 1-1e-8) -> logistic sample, clamped to [-1,1]."""
    idx = (logits - torch.log(-torch.log(u))).argmax(-1, keepdim=True).unsqueeze(-1)
    loc = torch.gather(locs, -1, idx).squeeze(-1)
    ls = torch.gather(log_scales, -1, idx).squeeze(-1)
    return (loc + torch.exp(ls) * (torch.log(u2) - torch.log(1 - u2))).clamp(-1, 1)


def srnn_audio_generate(sd, n_samples, max_timesteps, stack, eps, uniforms, num_mix=10, prefix="srnn"):
    """SRNNAudio.generate, unconditional (srnn.py:304-403, 515-535): d_t = GRU(encoder(x_{t-1}), d_{t-1}); z_t ~ prior(cat[d_t, z_{t-1}])
    (Elman transfer); x_t ~ DMoL(decoder(cat[z_t, d_t])) — always SAMPLED — fed back.  eps [T,B,z]; uniforms[t] = (u, u2).
    Returns x [B,T,stack,1]."""
    R = sd[f"{prefix}.d_forward_recurrent.weight_hh_l0"].size(1)
    Z = sd[f"{prefix}.prior.6.params.weight"].size(0) // 2
    g = f"{prefix}.d_forward_recurrent"
    x = torch.zeros(n_samples, stack)
    d_t, z_t = torch.zeros(n_samples, R), torch.zeros(n_samples, Z)
    out = []
    for t in range(max_timesteps):
        enc = _mlp(x, sd, f"{prefix}.encoder", (2, 4, 6), F.leaky_relu)
        d_t = gru_cell(enc, d_t, sd[f"{g}.weight_ih_l0"], sd[f"{g}.weight_hh_l0"], sd[f"{g}.bias_ih_l0"], sd[f"{g}.bias_hh_l0"])
        hp = _mlp(torch.cat([d_t, z_t], -1), sd, f"{prefix}.prior", (0, 2, 4), F.leaky_relu)
        mu_p, sd_p = gaussian_head(hp, sd[f"{prefix}.prior.6.params.weight"], sd[f"{prefix}.prior.6.params.bias"])
        z_t = eps[t] * sd_p + mu_p
        dec = _mlp(torch.cat([z_t, d_t], -1), sd, f"{prefix}.decoder", (0, 2, 4), F.leaky_relu).view(n_samples, stack, 3 * num_mix)
        logits, locs, log_scales = dmol_head(dec, sd[f"{prefix}.likelihood.params.weight"], sd[f"{prefix}.likelihood.params.bias"], num_mix)
        xs = dmol_sample(logits, locs, log_scales, *uniforms[t])  # [B,stack,1]
        out.append(xs)
        x = xs.squeeze(-1)
    return torch.stack(out, 1)


def wavenet_generate(sd, n_samples, n_frames, n_layers, n_stacks, uniforms, num_mix=10, base_dilation=2):
    """WaveNet.generate (wavenet.py:254-293), float input, n_stack_frames=1: per frame the stack runs over a receptive-field
    window, the single skip output is DIVIDED by sqrt(n_layers/n_stacks) (:274), ReLU-Linear-ReLU, DMoL head,
    sample (uniforms[t] = (u, u2)), FIFO shift.  Returns [B, n_frames, 1]."""
    dil = wavenet_dilations(n_layers, n_stacks, base_dilation)
    rf = sum(dil) + 1 + (sd["causal.conv.weight"].size(2) - 1)
    x = torch.zeros(n_samples, 1, rf)
    scale = math.sqrt(n_layers / n_stacks)
    out = []
    for t in range(n_frames):
        h = F.conv1d(x, sd["causal.conv.weight"], sd["causal.conv.bias"])
        skips = sum(residual_stack_skips(sd, "res_stack", h, dil, 1)) / scale  # [B,C,1]
        o = F.relu(F.linear(F.relu(skips.transpose(1, 2)), sd["out_transform.linear.weight"], sd["out_transform.linear.bias"]))
        logits, locs, log_scales = dmol_head(o, sd["likelihood.params.weight"], sd["likelihood.params.bias"], num_mix)
        pred = dmol_sample(logits, locs, log_scales, *uniforms[t])  # [B,1,1]
        out.append(pred)
        x = torch.cat([x[:, :, 1:], pred], dim=2)
    return torch.hstack(out)
