"""autograd glue over the C ABI: each Function enqueues hand-written gfx950 kernels (csrc/*.hip) on the current
torch HIP stream.  PyTorch is used for device memory, streams and autograd bookkeeping only — no ATen math on the
hot path, and no CPU fallback (``_hip.ptr`` refuses CPU tensors).

Layout convention: sequence tensors are time-major ``[T', B, F]``.
"""
from typing import List, Optional, Sequence

import torch

from . import _hip
from ._hip import VrnnWeights, check, load, ptr, stream_ptr

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
LEAKY_SLOPE = 0.01  # nn.LeakyReLU default (blvm/models/vrnn.py:491)

LAYOUT_BATCH_MAJOR, LAYOUT_TIME_MAJOR = 0, 1

# optional profiling hook: called with "fwd_begin"/"fwd_end"/"bwd_begin"/"bwd_end" around the recurrent-cell calls so
# a caller (bench.py) can record HIP events on the launching stream.  None = disabled.
seq_timer_hook = None


def _tick(tag):
    if seq_timer_hook is not None:
        seq_timer_hook(tag)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _hip.BlvmHipError(f"blvm HIP kernels compute in fp32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# raw launch helpers (no autograd)
# ----------------------------------------------------------------------------------------------------------------------


def gemm(op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, bias=None, act=ACT_NONE, slope=0.0, gate=None, ldg=0,
         accumulate=False, split_k=1):  # fmt: skip
    check(
        load().blvm_gemm_f32(op_a, op_b, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(C), ldc, ptr(bias), act, slope,
                             ptr(gate), ldg, int(accumulate), split_k, stream_ptr()),
        "blvm_gemm_f32",
    )  # fmt: skip


def colsum(X2d: torch.Tensor, out: torch.Tensor, accumulate=False):
    M, N = X2d.shape
    check(load().blvm_colsum_f32(M, N, ptr(X2d), X2d.stride(0), ptr(out), int(accumulate), stream_ptr()), "blvm_colsum_f32")


def upload_i32(host, device):
    """Small host integer tensor -> int32 device tensor without blocking the host (the values travel in kernel arguments)."""
    h = host.detach().to(device="cpu", dtype=torch.int32).contiguous()
    out = torch.empty(h.shape, device=device, dtype=torch.int32)
    if h.numel():
        check(load().blvm_upload_i32(h.data_ptr(), h.numel(), ptr(out), stream_ptr()), "blvm_upload_i32")
    return out


def wgrad_group(jobs, rows: int):
    """Weight (+ bias) gradients over the same `rows` as ONE launch (blvm_wgrad_group_f32): jobs = [(D [rows, N], X [rows, K], dW [N, K]
    or None, db [N] or None), ...], each dW += D^T X, db += D.sum(0); D and X may be row-strided views."""
    import ctypes

    jobs = [j for j in jobs if j[2] is not None or j[3] is not None]
    if not jobs:
        return
    n = len(jobs)
    ints = lambda v: (ctypes.c_int * n)(*v)  # noqa: E731

    def row_ptr(t):  # (row-strided views are fine: the leading dimension travels beside the pointer)
        if t is None:
            return None
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise _hip.BlvmHipError("wgrad_group: operands must be float32 [rows, cols] HIP tensors with unit column stride")
        return t.data_ptr()

    ptrs = lambda v: (ctypes.c_void_p * n)(*[ptr(t) for t in v])  # noqa: E731
    rptrs = lambda v: (ctypes.c_void_p * n)(*[row_ptr(t) for t in v])  # noqa: E731
    for D, X, dW, db in jobs:
        if D.shape[0] != rows or X.shape[0] != rows:
            raise _hip.BlvmHipError(f"wgrad_group: operands must be [rows={rows}, *], got {tuple(D.shape)} / {tuple(X.shape)}")
    check(
        load().blvm_wgrad_group_f32(n, ints([j[0].shape[1] for j in jobs]), ints([j[1].shape[1] for j in jobs]), rows,
                                    rptrs([j[0] for j in jobs]), ints([j[0].stride(0) for j in jobs]), rptrs([j[1] for j in jobs]),
                                    ints([j[1].stride(0) for j in jobs]), ptrs([j[2] for j in jobs]),
                                    ints([j[2].stride(0) if j[2] is not None else j[1].shape[1] for j in jobs]), ptrs([j[3] for j in jobs]),
                                    stream_ptr()),
        "blvm_wgrad_group_f32",
    )  # fmt: skip


def _zeros_like_many(tensors):
    """Zero-initialised gradient buffers for `tensors` carved out of ONE allocation (one fill launch instead of one per
    tensor; offsets kept 16-byte aligned)."""
    sizes = [(t.numel() + 3) // 4 * 4 for t in tensors]
    flat = torch.zeros(sum(sizes), device=tensors[0].device, dtype=torch.float32)
    out, off = [], 0
    for t, n in zip(tensors, sizes):
        out.append(flat[off : off + t.numel()].view(t.shape))
        off += n
    return out


def _pick_split(M, N, K):
    """Split-K factor of a weight-gradient GEMM [M,N] += A^T B over K rows: enough workgroups to fill 256 CUs."""
    if K >= 65536 and M >= 128 and N >= 128:  # conv-coder wgrads: K = L*B ~ 4e5 rows, 128x128 tiles, ~4 workgroups per CU
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        return max(1, min((1024 + tiles - 1) // tiles, K // 2048))
    bm = 64  # gemm_f32 cuts weight-gradient GEMMs into 64x64 tiles (csrc/common.h: gemm_pick_split)
    tiles = ((M + bm - 1) // bm) * ((N + bm - 1) // bm)
    s = (768 + tiles - 1) // tiles
    return max(1, min(s, (K + 255) // 256))


# ----------------------------------------------------------------------------------------------------------------------
# K6: MLP = chain of Linear + (Leaky)ReLU, one autograd node for the whole chain
# ----------------------------------------------------------------------------------------------------------------------


class _MLPFunction(torch.autograd.Function):
    """y = act(...act(x W1^T + b1)... Wn^T + bn) for a 2-D x; activation after EVERY layer (the reference's
    encoders/decoders end in an activation)."""

    @staticmethod
    def forward(ctx, x, act, slope, *params):
        x = _f32c(x)
        n_layers = len(params) // 2
        acts = [x]
        for l in range(n_layers):
            W, b = params[2 * l], params[2 * l + 1]
            inp = acts[-1]
            M, K = inp.shape
            N = W.shape[0]
            out = torch.empty(M, N, device=x.device, dtype=torch.float32)
            gemm(0, 0, M, N, K, inp, inp.stride(0), _f32c(W), K, out, N, bias=_f32c(b) if b is not None else None, act=act,
                 slope=slope)
            acts.append(out)
        ctx.act, ctx.slope, ctx.n_layers = act, slope, n_layers
        ctx.save_for_backward(*acts, *params)
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n_layers
        saved = ctx.saved_tensors
        acts, params = saved[: n + 1], saved[n + 1 :]
        slope = ctx.slope if ctx.act == ACT_LEAKY else 0.0
        dy = _f32c(dy)
        # derivative of the last activation (no producing GEMM to fuse it into)
        if ctx.act != ACT_NONE:
            dz = torch.empty_like(dy)
            check(load().blvm_act_bwd_f32(ptr(dy), ptr(acts[n]), slope, ptr(dz), dz.numel(), stream_ptr()), "blvm_act_bwd_f32")
        else:
            dz = dy
        grads: List[Optional[torch.Tensor]] = [None] * (2 * n)
        # every weight / bias gradient of the chain as a view of ONE zero-filled buffer (one fill launch instead of 2n)
        want = [ctx.needs_input_grad[3 + i] and params[i] is not None for i in range(2 * n)]
        sizes = [(params[i].numel() + 3) // 4 * 4 if want[i] else 0 for i in range(2 * n)]
        flat = torch.zeros(sum(sizes), device=dy.device, dtype=torch.float32)
        views, off = [], 0
        for i in range(2 * n):
            views.append(flat[off : off + params[i].numel()].view(params[i].shape) if want[i] else None)
            off += sizes[i]
        jobs = []  # every layer's weight + bias gradient: ONE grouped launch after the chain of dgrads (all share the rows)
        for l in range(n - 1, -1, -1):
            W, b = params[2 * l], params[2 * l + 1]
            inp = acts[l]
            M, K = inp.shape
            N = W.shape[0]
            # weight and bias gradient in one launch (the GEMM's first column block also sums the dz tiles it stages)
            dW, db = views[2 * l], views[2 * l + 1]
            jobs.append((dz, inp, dW, db))  # (keeps this layer's dz alive until the grouped launch below)
            grads[2 * l], grads[2 * l + 1] = dW, db
            if l > 0 or ctx.needs_input_grad[0]:
                dx = torch.empty(M, K, device=dz.device, dtype=torch.float32)
                # dgrad with the derivative of the PREVIOUS layer's activation fused into the epilogue
                gate = acts[l] if (l > 0 and ctx.act != ACT_NONE) else None
                gemm(0, 1, M, K, N, dz, N, _f32c(W), K, dx, K, slope=slope, gate=gate, ldg=K)
                dz = dx
        wgrad_group(jobs, acts[0].shape[0])
        return (dz if ctx.needs_input_grad[0] else None, None, None, *grads)


def mlp(x2d: torch.Tensor, layers: Sequence[torch.nn.Linear], act: int = ACT_LEAKY, slope: float = LEAKY_SLOPE):
    params = []
    for lin in layers:
        params += [lin.weight, lin.bias]
    return _MLPFunction.apply(x2d, act, slope, *params)


def linear(x2d: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, act: int = ACT_NONE, slope: float = 0.0):
    """act(x W^T + b) for a raw [out,in] weight (e.g. a 1x1 Conv1d weight viewed as a matrix)."""
    return _MLPFunction.apply(x2d, act, slope, weight, bias)


# ----------------------------------------------------------------------------------------------------------------------
# K7: DMoL head (Linear + log-likelihood + masked per-utterance sums)
# ----------------------------------------------------------------------------------------------------------------------


def _dmol_check_shapes(dec, W, y, B, T, Tp, S, num_mix):
    """The fused head is the per-frame [F,F] Linear of the stacked-frame models (F = 3 * num_mix); a head with another
    input width (e.g. CW-VAE's h -> F) runs as a K6 GEMM in front and the kernel gets W = None."""
    F = 3 * num_mix
    if dec.dim() != 2 or dec.shape[0] != B * Tp or dec.shape[1] != S * F:
        raise _hip.BlvmHipError(f"DMoL: activations must be [B*T'={B * Tp}, S*3*num_mix={S * F}], got {tuple(dec.shape)}")
    if W is not None and tuple(W.shape) != (F, F):
        raise _hip.BlvmHipError(f"DMoL: fused head weight must be [{F},{F}], got {tuple(W.shape)}")
    if tuple(y.shape) != (B, T) or Tp * S < T:
        raise _hip.BlvmHipError(f"DMoL: targets must be [B={B}, T={T}] with T <= T'*S={Tp * S}, got {tuple(y.shape)}")


def _head_linear_grads(d_par, dec, W, b, F, n_frames, need_w, need_b):
    """Gradients of a likelihood head's per-frame Linear(F, F): dW = d_par^T dec and db = column sums of d_par over all frames, in one
    launch when both are wanted (the bias gradient rides in the weight-gradient GEMM)."""
    dW = db = None
    if need_w and F % 4 != 0 and F % 2 == 0 and n_frames % 2 == 0 and n_frames >= 4096:
        # F = 30: rows of 120 bytes are not 16-byte aligned and a 30 x 30 output uses a fifth of the GEMM's 64 x 64 tile (measured:
        # 147 us for 246 MB at [64,16000], scalar loads).  Two frames per row -> [n/2, 2F] operands with aligned 240-byte rows and a
        # 60 x 60 product over half the rows whose two diagonal blocks are the wanted sums (even frames | odd frames).
        T2 = torch.zeros(2 * F, 2 * F, device=W.device, dtype=torch.float32)
        tb = torch.zeros(2 * F, device=W.device, dtype=torch.float32) if need_b else None
        half = n_frames // 2
        check(load().blvm_wgrad_f32(2 * F, 2 * F, half, ptr(d_par), 2 * F, ptr(dec), 2 * F, ptr(T2), T2.stride(0), ptr(tb),
                                    max(128, min(512, half // 1024)), stream_ptr()), "blvm_wgrad_f32")  # fmt: skip
        dW = T2[:F, :F] + T2[F : 2 * F, F : 2 * F]
        db = tb[:F] + tb[F:] if need_b else None
        return dW, db
    if need_w:
        dW = torch.zeros_like(W)
        db = torch.zeros_like(b) if need_b else None
        check(load().blvm_wgrad_f32(F, F, n_frames, ptr(d_par), F, ptr(dec), F, ptr(dW), F, ptr(db), max(256, min(1024, n_frames // 1024)),
                                    stream_ptr()), "blvm_wgrad_f32")  # fmt: skip
    elif need_b:
        db = torch.empty_like(b)
        colsum(d_par.view(n_frames, F), db)
    return dW, db


class _DMoLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, num_mix, num_bins, log_eps):
        dec, y = _f32c(dec), _f32c(y)
        W, b = (_f32c(W), _f32c(b)) if W is not None else (None, None)
        _dmol_check_shapes(dec, W, y, B, T, Tp, S, num_mix)
        log_prob = torch.zeros(B, device=dec.device, dtype=torch.float64)
        check(
            load().blvm_dmol_fwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), B, T, Tp, S, num_mix, num_bins,
                                 log_eps, ptr(log_prob), None, stream_ptr()),
            "blvm_dmol_fwd",
        )  # fmt: skip
        ctx.has_linear = W is not None
        ctx.save_for_backward(dec, y, x_sl_dev, *((W, b) if W is not None else ()))
        ctx.cfg = (layout, B, T, Tp, S, num_mix, num_bins, log_eps)
        return log_prob

    @staticmethod
    def backward(ctx, g):
        dec, y, x_sl_dev, *lin = ctx.saved_tensors
        W, b = lin if ctx.has_linear else (None, None)
        layout, B, T, Tp, S, num_mix, num_bins, log_eps = ctx.cfg
        g32 = g.to(torch.float32).contiguous()
        F = 3 * num_mix
        d_dec = torch.empty_like(dec)
        d_par = torch.empty_like(dec) if ctx.has_linear else None
        check(
            load().blvm_dmol_bwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), ptr(g32), B, T, Tp, S, num_mix,
                                 num_bins, log_eps, ptr(d_dec), ptr(d_par), stream_ptr()),
            "blvm_dmol_bwd",
        )  # fmt: skip
        n_frames = dec.numel() // F
        dW, db = _head_linear_grads(d_par, dec, W, b, F, n_frames, ctx.has_linear and ctx.needs_input_grad[1], ctx.has_linear and ctx.needs_input_grad[2])
        return (d_dec if ctx.needs_input_grad[0] else None, dW, db) + (None,) * 10


def dmol_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, num_mix=10, num_bins=256, log_eps=-7.0):
    """Per-utterance masked DMoL log-likelihood sums [B] (float64) of targets y [B,T] given decoder activations
    `dec` ([rows, S*3*num_mix], rows ordered by `layout`)."""
    return _DMoLFunction.apply(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, num_mix, num_bins, log_eps)


class _GaussHeadFunction(torch.autograd.Function):
    """K7b (kind 1, Gaussian mixture, F = 3*num_mix) / K7c (kind 2, single Gaussian, F = 2): decoder activations -> per-frame
    Linear(F->F) (optional) -> log-likelihood -> masked per-utterance float64 sums [B]."""

    @staticmethod
    def forward(ctx, dec, W, b, y, x_sl_dev, kind, layout, B, T, Tp, S, num_mix, sd_beta, sd_eps):
        dec, y = _f32c(dec), _f32c(y)
        W, b = (_f32c(W), _f32c(b)) if W is not None else (None, None)
        F = 3 * num_mix if kind == 1 else 2
        if dec.dim() != 2 or dec.shape[0] != B * Tp or dec.shape[1] != S * F:
            raise _hip.BlvmHipError(f"Gaussian head: activations must be [B*T'={B * Tp}, S*F={S * F}], got {tuple(dec.shape)}")
        if W is not None and tuple(W.shape) != (F, F):
            raise _hip.BlvmHipError(f"Gaussian head: fused head weight must be [{F},{F}], got {tuple(W.shape)}")
        if tuple(y.shape) != (B, T) or Tp * S < T:
            raise _hip.BlvmHipError(f"Gaussian head: targets must be [B={B}, T={T}] with T <= T'*S={Tp * S}, got {tuple(y.shape)}")
        log_prob = torch.zeros(B, device=dec.device, dtype=torch.float64)
        lib = load()
        if kind == 1:
            rc = lib.blvm_gmm_fwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), B, T, Tp, S, num_mix, sd_beta, sd_eps,
                                  ptr(log_prob), None, stream_ptr())  # fmt: skip
        else:
            rc = lib.blvm_gauss_head_fwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), B, T, Tp, S, sd_beta, sd_eps,
                                         ptr(log_prob), None, stream_ptr())  # fmt: skip
        check(rc, "blvm_gmm_fwd" if kind == 1 else "blvm_gauss_head_fwd")
        ctx.has_linear = W is not None
        ctx.save_for_backward(dec, y, x_sl_dev, *((W, b) if W is not None else ()))
        ctx.cfg = (kind, layout, B, T, Tp, S, num_mix, sd_beta, sd_eps, F)
        return log_prob

    @staticmethod
    def backward(ctx, g):
        dec, y, x_sl_dev, *lin = ctx.saved_tensors
        W, b = lin if ctx.has_linear else (None, None)
        kind, layout, B, T, Tp, S, num_mix, sd_beta, sd_eps, F = ctx.cfg
        g32 = g.to(torch.float32).contiguous()
        d_dec = torch.empty_like(dec)
        d_par = torch.empty_like(dec) if ctx.has_linear else None
        lib = load()
        if kind == 1:
            rc = lib.blvm_gmm_bwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), ptr(g32), B, T, Tp, S, num_mix, sd_beta,
                                  sd_eps, ptr(d_dec), ptr(d_par), stream_ptr())  # fmt: skip
        else:
            rc = lib.blvm_gauss_head_bwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), ptr(g32), B, T, Tp, S, sd_beta,
                                         sd_eps, ptr(d_dec), ptr(d_par), stream_ptr())  # fmt: skip
        check(rc, "blvm_gmm_bwd" if kind == 1 else "blvm_gauss_head_bwd")
        n_frames = dec.numel() // F
        dW, db = _head_linear_grads(d_par, dec, W, b, F, n_frames, ctx.has_linear and ctx.needs_input_grad[1], ctx.has_linear and ctx.needs_input_grad[2])
        return (d_dec if ctx.needs_input_grad[0] else None, dW, db) + (None,) * 11


def gmm_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, num_mix, sd_beta, sd_eps):
    """Per-utterance masked Gaussian-mixture log-likelihood sums [B] (float64); dec [rows, S*3*num_mix]."""
    return _GaussHeadFunction.apply(dec, W, b, y, x_sl_dev, 1, layout, B, T, Tp, S, num_mix, float(sd_beta), float(sd_eps))


def gauss_log_prob(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, sd_beta, sd_eps):
    """Per-utterance masked Gaussian log-likelihood sums [B] (float64); dec [rows, S*2]."""
    return _GaussHeadFunction.apply(dec, W, b, y, x_sl_dev, 2, layout, B, T, Tp, S, 10, float(sd_beta), float(sd_eps))


def mix_sample(par, u, v, kind: int, log_eps: float = -7.0, sd_beta: float = 1.0, sd_eps: float = 0.0):
    """Sample / mode of a 10-component mixture head (no autograd): par [..., 30] head outputs, u [..., 10] uniforms or None (mode),
    v [...] component noise or None (location).  kind 0 DMoL, 1 GMM.  Returns [...]."""
    par = _f32c(par)
    lead = par.shape[:-1]
    n = par.numel() // par.shape[-1]
    out = torch.empty(n, device=par.device, dtype=torch.float32)
    u = _f32c(u.reshape(n, -1)) if u is not None else None
    v = _f32c(v.reshape(n)) if v is not None else None
    check(load().blvm_mix_sample(ptr(par), ptr(u), ptr(v), n, par.shape[-1] // 3, kind, log_eps, sd_beta, sd_eps, ptr(out), stream_ptr()),
          "blvm_mix_sample")  # fmt: skip
    return out.view(*lead)


def dmol_ll_twise(dec, W, b, y, x_sl_dev, layout, B, T, Tp, S, num_mix=10, num_bins=256, log_eps=-7.0):
    """Masked frame-wise log-likelihood [B,T] (no autograd)."""
    dec, y = _f32c(dec), _f32c(y)
    W, b = (_f32c(W), _f32c(b)) if W is not None else (None, None)
    _dmol_check_shapes(dec, W, y, B, T, Tp, S, num_mix)
    lp = torch.zeros(B, device=dec.device, dtype=torch.float64)
    ll = torch.zeros(B, T, device=dec.device, dtype=torch.float32)
    check(
        load().blvm_dmol_fwd(ptr(dec), layout, ptr(W), ptr(b), ptr(y), ptr(x_sl_dev), B, T, Tp, S, num_mix, num_bins,
                             log_eps, ptr(lp), ptr(ll), stream_ptr()),
        "blvm_dmol_fwd",
    )  # fmt: skip
    return ll, lp


# ----------------------------------------------------------------------------------------------------------------------
# K8: Gaussian KL + free nats + masked sums (stand-alone form; the VRNN path fuses the backward into K1)
# ----------------------------------------------------------------------------------------------------------------------


class _KLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu_q, sd_q, mu_p, sd_p, x_sl_dev, layout, B, Tp, Z, stride, fn_floor):
        mu_q, sd_q, mu_p, sd_p = (_f32c(t) for t in (mu_q, sd_q, mu_p, sd_p))
        kld = torch.zeros(B, device=mu_q.device, dtype=torch.float64)
        kld_fn = torch.zeros(B, device=mu_q.device, dtype=torch.float64)
        check(
            load().blvm_kl_fwd(ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), layout, ptr(x_sl_dev), B, Tp, Z, stride,
                               fn_floor, ptr(kld), ptr(kld_fn), stream_ptr()),
            "blvm_kl_fwd",
        )  # fmt: skip
        ctx.save_for_backward(mu_q, sd_q, mu_p, sd_p, x_sl_dev)
        ctx.cfg = (layout, B, Tp, Z, stride, fn_floor)
        return kld, kld_fn

    @staticmethod
    def backward(ctx, g_raw, g_fn):
        mu_q, sd_q, mu_p, sd_p, x_sl_dev = ctx.saved_tensors
        layout, B, Tp, Z, stride, fn_floor = ctx.cfg
        c_raw = g_raw.to(torch.float32).contiguous()
        c_fn = g_fn.to(torch.float32).contiguous()
        outs = [torch.empty_like(mu_q) for _ in range(4)]
        check(
            load().blvm_kl_bwd(ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), layout, ptr(x_sl_dev), ptr(c_raw), ptr(c_fn),
                               B, Tp, Z, stride, fn_floor, *(ptr(o) for o in outs), stream_ptr()),
            "blvm_kl_bwd",
        )  # fmt: skip
        return (*outs, None, None, None, None, None, None, None)


def gaussian_kl_sums(mu_q, sd_q, mu_p, sd_p, x_sl_dev, layout, B, Tp, Z, stride, free_nats=0.0):
    """(kld[B], kld_fn[B]) in float64: masked sums of the analytic KL and of max(KL, free_nats/Z)."""
    fn_floor = float(free_nats) / Z if free_nats else 0.0
    return _KLFunction.apply(mu_q, sd_q, mu_p, sd_p, x_sl_dev, layout, B, Tp, Z, stride, fn_floor)


# ----------------------------------------------------------------------------------------------------------------------
# K1: VRNN cell over a sequence
# ----------------------------------------------------------------------------------------------------------------------

_VRNN_PARAM_ORDER = (
    ["prior_w0", "prior_b0", "prior_w1", "prior_b1", "prior_w2", "prior_b2", "prior_hw", "prior_hb"]
    + ["post_w0", "post_b0", "post_w1", "post_b1", "post_w2", "post_b2", "post_hw", "post_hb"]
    + ["phi_w0", "phi_b0", "phi_w1", "phi_b1", "phi_w2", "phi_b2", "phi_w3", "phi_b3"]
    + ["gru_wih", "gru_whh", "gru_bih", "gru_bhh"]
)


class _GaussLatentFunction(torch.autograd.Function):
    """(mu_p, sd_p_raw, mu_q_raw, sd_q_raw, eps) -> (sd_p, mu_q, sd_q, z): softplus heads, posterior combination and the
    reparameterised sample of one STCN latent level, elementwise (K8b)."""

    @staticmethod
    def forward(ctx, mu_p, sp_raw, mq, sq_raw, eps, cfg):
        mu_p, sp_raw, mq, sq_raw, eps = (_f32c(t) for t in (mu_p, sp_raw, mq, sq_raw, eps))
        beta_p, beta_q, sd_eps, mode = cfg
        outs = [torch.empty_like(mu_p) for _ in range(4)]
        check(load().blvm_gauss_latent_fwd(ptr(mu_p), ptr(sp_raw), ptr(mq), ptr(sq_raw), ptr(eps), mu_p.numel(), beta_p, beta_q,
                                           sd_eps, mode, *(ptr(o) for o in outs), stream_ptr()), "blvm_gauss_latent_fwd")  # fmt: skip
        ctx.cfg = cfg
        ctx.save_for_backward(mu_p, sp_raw, mq, sq_raw, eps)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_sd_p, g_mu_q, g_sd_q, g_z):
        mu_p, sp_raw, mq, sq_raw, eps = ctx.saved_tensors
        beta_p, beta_q, sd_eps, mode = ctx.cfg
        gs = [_f32c(g) if g is not None else None for g in (g_sd_p, g_mu_q, g_sd_q, g_z)]
        outs = [torch.empty_like(mu_p) for _ in range(4)]
        check(load().blvm_gauss_latent_bwd(ptr(mu_p), ptr(sp_raw), ptr(mq), ptr(sq_raw), ptr(eps), *(ptr(g) for g in gs),
                                           mu_p.numel(), beta_p, beta_q, sd_eps, mode, *(ptr(o) for o in outs), stream_ptr()),
              "blvm_gauss_latent_bwd")  # fmt: skip
        return (*outs, None, None)


def gauss_latent(mu_p, sd_p_raw, mu_q_raw, sd_q_raw, eps, beta_p: float, beta_q: float, sd_eps: float, mode: int):
    return _GaussLatentFunction.apply(mu_p, sd_p_raw, mu_q_raw, sd_q_raw, eps, (float(beta_p), float(beta_q), float(sd_eps), int(mode)))


def _pack_weights(ts: Sequence[Optional[torch.Tensor]]) -> VrnnWeights:
    d = dict(zip(_VRNN_PARAM_ORDER, ts))
    p = lambda k: ptr(d[k]) if d[k] is not None else None  # noqa: E731
    w = VrnnWeights()
    for i in range(3):
        w.prior_w[i], w.prior_b[i] = p(f"prior_w{i}"), p(f"prior_b{i}")
        w.post_w[i], w.post_b[i] = p(f"post_w{i}"), p(f"post_b{i}")
    for i in range(4):
        w.phi_w[i], w.phi_b[i] = p(f"phi_w{i}"), p(f"phi_b{i}")
    w.prior_hw, w.prior_hb, w.post_hw, w.post_hb = p("prior_hw"), p("prior_hb"), p("post_hw"), p("post_hb")
    w.gru_wih, w.gru_whh, w.gru_bih, w.gru_bhh = p("gru_wih"), p("gru_whh"), p("gru_bih"), p("gru_bhh")
    return w


@torch.no_grad()
def vrnn_decode(enc_lin, cell_params, dec_lin, lik_lin, x0, h0, eps, u, v, S, H, Z, R, num_mix, sd_eps, slope, log_eps, whole_chip=None):
    """K1c: T = eps.shape[0] steps of ancestral sampling for all B utterances in one launch.  enc_lin / dec_lin: 3 nn.Linear each;
    cell_params in `_VRNN_PARAM_ORDER`; lik_lin the DMoL head's Linear.  x0 [B,S], h0 [B,R] or None, eps [T,B,Z],
    u [T,B,S,num_mix] / v [T,B,S] uniforms (None: the mode).  -> (x [B,T,S], h_n [B,R]).
    whole_chip: True = `blvm_vrnn_generate` (every layer of a step dealt over all CUs, B <= 128), False = `blvm_vrnn_decode`
    (16 utterances per CU), None = the former whenever it applies."""
    from ._hip import VrnnDecodeWeights

    lib = load()
    T, B = eps.shape[0], x0.shape[0]
    dev = x0.device
    keep = [_f32c(t) for lin in (*enc_lin, *dec_lin, lik_lin) for t in (lin.weight, lin.bias)]
    cp = [_f32c(p) for p in cell_params]
    cw = _pack_weights(cp)
    w = VrnnDecodeWeights()
    for i in range(3):
        w.enc_w[i], w.enc_b[i] = ptr(keep[2 * i]), ptr(keep[2 * i + 1])
        w.dec_w[i], w.dec_b[i] = ptr(keep[6 + 2 * i]), ptr(keep[6 + 2 * i + 1])
    w.lik_w, w.lik_b = ptr(keep[12]), ptr(keep[13])
    import ctypes

    w.cell = ctypes.pointer(cw)
    x0, eps = _f32c(x0), _f32c(eps)
    h0 = _f32c(h0) if h0 is not None else None
    u = _f32c(u) if u is not None else None
    v = _f32c(v) if v is not None else None
    if whole_chip is None:
        whole_chip = B <= lib.blvm_pchain_max_batch()
    x = torch.empty(B, T, S, device=dev, dtype=torch.float32)
    hn = torch.empty(B, R, device=dev, dtype=torch.float32)
    if whole_chip:
        scratch = torch.empty(lib.blvm_vrnn_generate_scratch_floats(T, B, S, H, Z, R), device=dev, dtype=torch.float32)
        check(lib.blvm_vrnn_generate(ctypes.byref(w), ptr(x0), ptr(h0), ptr(eps), ptr(u), ptr(v), T, B, S, H, Z, R, num_mix, sd_eps,
                                     slope, log_eps, ptr(x), ptr(hn), ptr(scratch), stream_ptr()), "blvm_vrnn_generate")  # fmt: skip
        return x, hn
    scratch = torch.empty(lib.blvm_vrnn_decode_scratch_floats(S, H, Z, R), device=dev, dtype=torch.float32)
    check(lib.blvm_vrnn_decode(ctypes.byref(w), ptr(x0), ptr(h0), ptr(eps), ptr(u), ptr(v), T, B, S, H, Z, R, num_mix, sd_eps, slope,
                               log_eps, ptr(x), ptr(hn), ptr(scratch), stream_ptr()), "blvm_vrnn_decode")  # fmt: skip
    return x, hn


class _VRNNSeqFunction(torch.autograd.Function):
    """(enc, h0, eps, 28 params) -> decin [T'+1,B,H+R], kld [B], kld_fn [B]  (+ non-differentiable mu/sd/z)."""

    @staticmethod
    def forward(ctx, enc, h0, eps, x_sl_dev, cfg, *params):
        Tp, B, X, H, Z, R, residual, sd_eps, stride, fn_floor = cfg
        enc, eps = _f32c(enc), _f32c(eps)
        params = tuple(_f32c(p) for p in params)
        dev = enc.device
        lib = load()
        f32 = dict(device=dev, dtype=torch.float32)
        decin = torch.empty(Tp + 1, B, H + R, **f32)
        decin[Tp, :, :H].zero_()  # the phi-part of the extra row is never written by the kernels
        mu_q, sd_q, mu_p, sd_p, z = (torch.empty(Tp, B, Z, **f32) for _ in range(5))
        reserve = torch.empty(lib.blvm_vrnn_reserve_floats(Tp, B, X, H, Z, R), **f32)
        w = _pack_weights(params)
        _tick("fwd_begin")
        check(
            lib.blvm_vrnn_seq_fwd(w, ptr(enc), ptr(_f32c(h0)) if h0 is not None else None, ptr(eps), Tp, B, X, H, Z, R,
                                  int(residual), sd_eps, ptr(decin), ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), ptr(z),
                                  ptr(reserve), stream_ptr()),
            "blvm_vrnn_seq_fwd",
        )  # fmt: skip
        _tick("fwd_end")
        kld = torch.zeros(B, device=dev, dtype=torch.float64)
        kld_fn = torch.zeros(B, device=dev, dtype=torch.float64)
        check(
            lib.blvm_kl_fwd(ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), LAYOUT_TIME_MAJOR, ptr(x_sl_dev), B, Tp, Z, stride,
                            fn_floor, ptr(kld), ptr(kld_fn), stream_ptr()),
            "blvm_kl_fwd",
        )  # fmt: skip
        ctx.cfg = cfg
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(enc, eps, x_sl_dev, decin, mu_q, sd_q, mu_p, sd_p, z, reserve, *params)
        ctx.mark_non_differentiable(mu_q, sd_q, mu_p, sd_p, z)
        ctx.set_materialize_grads(False)  # (else autograd fills a [T',B,Z] zero gradient per statistics output: 16 MB each at [64,16000])
        return decin, kld, kld_fn, mu_q, sd_q, mu_p, sd_p, z

    @staticmethod
    def backward(ctx, d_decin, g_kld, g_kld_fn, *_unused):
        Tp, B, X, H, Z, R, residual, sd_eps, stride, fn_floor = ctx.cfg
        enc, eps, x_sl_dev, decin, mu_q, sd_q, mu_p, sd_p, z, reserve, *params = ctx.saved_tensors
        dev = enc.device
        lib = load()
        f32 = dict(device=dev, dtype=torch.float32)
        d_decin = _f32c(d_decin) if d_decin is not None else torch.zeros_like(decin)
        c_raw = g_kld.to(torch.float32).contiguous() if g_kld is not None else None
        c_fn = g_kld_fn.to(torch.float32).contiguous() if g_kld_fn is not None else None
        grads = _zeros_like_many(params)
        d_enc = torch.empty_like(enc)
        d_h0 = torch.empty(B, R, **f32) if ctx.has_h0 else None
        ws = torch.empty(lib.blvm_vrnn_bwd_workspace_floats(Tp, B, X, H, Z, R), **f32)
        w, g = _pack_weights(params), _pack_weights(grads)
        _tick("bwd_begin")
        check(
            lib.blvm_vrnn_seq_bwd(w, ptr(enc), ptr(eps), ptr(decin), ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), ptr(z),
                                  ptr(reserve), ptr(d_decin), ptr(x_sl_dev), ptr(c_raw), ptr(c_fn), stride, fn_floor, Tp, B,
                                  X, H, Z, R, int(residual), sd_eps, ptr(d_enc), ptr(d_h0), g, ptr(ws), stream_ptr()),
            "blvm_vrnn_seq_bwd",
        )  # fmt: skip
        _tick("bwd_end")
        return (d_enc, d_h0, None, None, None, *grads)


def vrnn_sequence(enc, h0, eps, x_sl_dev, params: Sequence[torch.Tensor], X, H, Z, R, residual_posterior, stride,
                  free_nats=0.0, sd_eps=1e-6):  # fmt: skip
    """Run the VRNN cell over enc [T',B,X].  `params` in `_VRNN_PARAM_ORDER`.  Returns
    (decin [T'+1,B,H+R], kld [B] f64, kld_fn [B] f64, mu_q, sd_q, mu_p, sd_p, z)."""
    Tp, B, _ = enc.shape
    fn_floor = float(free_nats) / Z if free_nats else 0.0
    cfg = (Tp, B, X, H, Z, R, int(residual_posterior), float(sd_eps), int(stride), fn_floor)  # 0 plain, 1 residual, 3 generate
    return _VRNNSeqFunction.apply(enc, h0, eps, x_sl_dev, cfg, *params)


# ----------------------------------------------------------------------------------------------------------------------
# K4: LSTM over a sequence (packed-sequence semantics through `lens`)
# ----------------------------------------------------------------------------------------------------------------------


class _LSTMSeqFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, h0, c0, lens_dev, Wih, Whh, bih, bhh):
        inp, Wih, Whh, bih, bhh = (_f32c(t) for t in (inp, Wih, Whh, bih, bhh))
        T, B, I = inp.shape
        H = Whh.shape[1]
        lib = load()
        f32 = dict(device=inp.device, dtype=torch.float32)
        out = torch.empty(T, B, H, **f32)
        hn, cn = torch.empty(B, H, **f32), torch.empty(B, H, **f32)
        reserve = torch.empty(lib.blvm_lstm_reserve_floats(T, B, H), **f32)
        check(
            lib.blvm_lstm_seq_fwd(ptr(Wih), ptr(Whh), ptr(bih), ptr(bhh), ptr(inp), ptr(_f32c(h0)) if h0 is not None else None,
                                  ptr(_f32c(c0)) if c0 is not None else None, ptr(lens_dev), T, B, I, H, ptr(out), ptr(hn), ptr(cn),
                                  ptr(reserve), stream_ptr()),
            "blvm_lstm_seq_fwd",
        )  # fmt: skip
        ctx.dims = (T, B, I, H)
        ctx.has_state = (h0 is not None, c0 is not None)
        ctx.save_for_backward(inp, Wih, Whh, reserve)
        ctx.mark_non_differentiable(hn, cn)
        return out, hn, cn

    @staticmethod
    def backward(ctx, d_out, _dhn, _dcn):
        T, B, I, H = ctx.dims
        inp, Wih, Whh, reserve = ctx.saved_tensors
        lib = load()
        f32 = dict(device=inp.device, dtype=torch.float32)
        d_out = _f32c(d_out)
        d_in = torch.empty_like(inp) if ctx.needs_input_grad[0] else None
        d_h0 = torch.empty(B, H, **f32) if ctx.has_state[0] and ctx.needs_input_grad[1] else None
        d_c0 = torch.empty(B, H, **f32) if ctx.has_state[1] and ctx.needs_input_grad[2] else None
        dWih, dWhh = torch.zeros_like(Wih), torch.zeros_like(Whh)
        dbih, dbhh = torch.zeros(4 * H, **f32), torch.zeros(4 * H, **f32)
        ws = torch.empty(lib.blvm_lstm_bwd_workspace_floats(T, B, H), **f32)
        check(
            lib.blvm_lstm_seq_bwd(ptr(Wih), ptr(Whh), ptr(inp), ptr(reserve), ptr(d_out), T, B, I, H, ptr(d_in), ptr(d_h0),
                                  ptr(d_c0), ptr(dWih), ptr(dWhh), ptr(dbih), ptr(dbhh), ptr(ws), stream_ptr()),
            "blvm_lstm_seq_bwd",
        )  # fmt: skip
        return d_in, d_h0, d_c0, None, dWih, dWhh, dbih, dbhh


def lstm_sequence(inp, h0, c0, lens_dev, Wih, Whh, bih, bhh):
    """inp [T,B,I] -> (out [T,B,H] zero past each row's length, h_n, c_n [B,H])."""
    return _LSTMSeqFunction.apply(inp, h0, c0, lens_dev, Wih, Whh, bih, bhh)


# ----------------------------------------------------------------------------------------------------------------------
# K2: GRU over a sequence, optionally reversed per row (reverse_sequences folded into the kernel's index map)
# ----------------------------------------------------------------------------------------------------------------------


class _GRUSeqFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, h0, lens_dev, reverse, Wih, Whh, bih, bhh):
        inp, Wih, Whh, bih, bhh = (_f32c(t) for t in (inp, Wih, Whh, bih, bhh))
        T, B, I = inp.shape
        R = Whh.shape[1]
        lib = load()
        f32 = dict(device=inp.device, dtype=torch.float32)
        out = torch.empty(T, B, R, **f32)
        hn = torch.empty(B, R, **f32)
        reserve = torch.empty(lib.blvm_gru_reserve_floats(T, B, R), **f32)
        check(
            lib.blvm_gru_seq_fwd(ptr(Wih), ptr(Whh), ptr(bih), ptr(bhh), ptr(inp), I, ptr(_f32c(h0)) if h0 is not None else None,
                                 ptr(lens_dev), int(reverse), T, B, I, R, ptr(out), B * R, R, ptr(hn), ptr(reserve), stream_ptr()),
            "blvm_gru_seq_fwd",
        )  # fmt: skip
        ctx.dims = (T, B, I, R, int(reverse))
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(inp, Wih, Whh, reserve, lens_dev)
        ctx.mark_non_differentiable(hn)
        return out, hn

    @staticmethod
    def backward(ctx, d_out, _dhn):
        T, B, I, R, reverse = ctx.dims
        inp, Wih, Whh, reserve, lens_dev = ctx.saved_tensors
        lib = load()
        f32 = dict(device=inp.device, dtype=torch.float32)
        d_out = _f32c(d_out)
        d_in = torch.empty_like(inp) if ctx.needs_input_grad[0] else None
        d_h0 = torch.empty(B, R, **f32) if ctx.has_h0 and ctx.needs_input_grad[1] else None
        dWih, dWhh = torch.zeros_like(Wih), torch.zeros_like(Whh)
        dbih, dbhh = torch.zeros(3 * R, **f32), torch.zeros(3 * R, **f32)
        ws = torch.empty(lib.blvm_gru_bwd_workspace_floats(T, B, R), **f32)
        check(
            lib.blvm_gru_seq_bwd(ptr(Wih), ptr(Whh), ptr(inp), I, ptr(lens_dev), reverse, ptr(reserve), ptr(d_out), B * R, R, T, B,
                                 I, R, ptr(d_in), I, 0, ptr(d_h0), ptr(dWih), ptr(dWhh), ptr(dbih), ptr(dbhh), ptr(ws),
                                 stream_ptr()),
            "blvm_gru_seq_bwd",
        )  # fmt: skip
        return d_in, d_h0, None, None, dWih, dWhh, dbih, dbhh


def gru_sequence(inp, h0, Wih, Whh, bih, bhh, lens_dev=None, reverse=False):
    """inp [T,B,I] -> (out [T,B,R], h_n [B,R]).  reverse=True: per-row time reversal that leaves right padding in
    place (needs lens_dev [B] int32) — equivalent to reverse_sequences -> GRU -> reverse_sequences."""
    return _GRUSeqFunction.apply(inp, h0, lens_dev, reverse, Wih, Whh, bih, bhh)


# ----------------------------------------------------------------------------------------------------------------------
# K3: SRNN latent chain
# ----------------------------------------------------------------------------------------------------------------------

_SRNN_PARAM_ORDER = (
    ["prior_w0", "prior_b0", "prior_w1", "prior_b1", "prior_w2", "prior_b2", "prior_hw", "prior_hb"]
    + ["post_w0", "post_b0", "post_w1", "post_b1", "post_w2", "post_b2", "post_hw", "post_hb"]
)


def _pack_srnn(ts):
    d = dict(zip(_SRNN_PARAM_ORDER, ts))
    w = _hip.SrnnWeights()
    for i in range(3):
        w.prior_w[i], w.prior_b[i] = ptr(d[f"prior_w{i}"]), ptr(d[f"prior_b{i}"])
        w.post_w[i], w.post_b[i] = ptr(d[f"post_w{i}"]), ptr(d[f"post_b{i}"])
    w.prior_hw, w.prior_hb, w.post_hw, w.post_hb = ptr(d["prior_hw"]), ptr(d["prior_hb"]), ptr(d["post_hw"]), ptr(d["post_hb"])
    return w


def srnn_generate(enc_lin, gru, chain_params, dec_lin, lik_lin, x0, d0, z0, eps, u, v, S, H, Z, R, num_mix, sd_eps, slope, log_eps):
    """K3c: T = eps.shape[0] steps of ancestral sampling from SRNNAudio for all B <= 128 utterances in one persistent launch.
    enc_lin / dec_lin: 3 nn.Linear each; gru: the forward nn.GRU (parameter container); chain_params in `_SRNN_PARAM_ORDER`;
    lik_lin the DMoL head's Linear.  x0 [B,S]; d0 [B,R], z0 [B,Z] or None; eps [T,B,Z]; u [T,B,S,num_mix] / v [T,B,S] (None: the
    mode).  -> (x [B,T,S], d_T [B,R], z [T,B,Z])."""
    import ctypes

    lib = load()
    T, B = eps.shape[0], x0.shape[0]
    dev = x0.device
    keep = [_f32c(t) for lin in (*enc_lin, *dec_lin, lik_lin) for t in (lin.weight, lin.bias)]
    gk = [_f32c(t) for t in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
    cp = [_f32c(p) for p in chain_params]
    cw = _pack_srnn(cp)
    w = _hip.SrnnDecodeWeights()
    for i in range(3):
        w.enc_w[i], w.enc_b[i] = ptr(keep[2 * i]), ptr(keep[2 * i + 1])
        w.dec_w[i], w.dec_b[i] = ptr(keep[6 + 2 * i]), ptr(keep[6 + 2 * i + 1])
    w.lik_w, w.lik_b = ptr(keep[12]), ptr(keep[13])
    w.gru_wih, w.gru_whh, w.gru_bih, w.gru_bhh = (ptr(t) for t in gk)
    w.chain = ctypes.pointer(cw)
    x0, eps = _f32c(x0), _f32c(eps)
    d0 = _f32c(d0) if d0 is not None else None
    z0 = _f32c(z0) if z0 is not None else None
    u = _f32c(u) if u is not None else None
    v = _f32c(v) if v is not None else None
    f32 = dict(device=dev, dtype=torch.float32)
    scratch = torch.empty(lib.blvm_srnn_generate_scratch_floats(T, B, S, H, Z, R), **f32)
    x, dn, zs = torch.empty(B, T, S, **f32), torch.empty(B, R, **f32), torch.empty(T, B, Z, **f32)
    check(lib.blvm_srnn_generate(ctypes.byref(w), ptr(x0), ptr(d0), ptr(z0), ptr(eps), ptr(u), ptr(v), T, B, S, H, Z, R, num_mix, sd_eps,
                                 slope, log_eps, ptr(x), ptr(dn), ptr(zs), ptr(scratch), stream_ptr()), "blvm_srnn_generate")  # fmt: skip
    return x, dn, zs


class _SRNNLatentFunction(torch.autograd.Function):
    """(d, a, z0, eps, 16 params) -> zs [T'+1,B,Z], kld [B], kld_fn [B]  (+ non-differentiable mu/sd)."""

    @staticmethod
    def forward(ctx, d, a, z0, eps, x_sl_dev, cfg, *params):
        Tp, B, H, Z, R, residual, sd_eps, slope, stride, fn_floor = cfg
        d, a, eps = _f32c(d), _f32c(a), _f32c(eps)
        params = tuple(_f32c(p) for p in params)
        lib = load()
        f32 = dict(device=d.device, dtype=torch.float32)
        zs = torch.empty(Tp + 1, B, Z, **f32)
        mu_q, sd_q, mu_p, sd_p = (torch.empty(Tp, B, Z, **f32) for _ in range(4))
        reserve = torch.empty(lib.blvm_srnn_reserve_floats(Tp, B, H, Z, R), **f32)
        _tick("fwd_begin")
        check(
            lib.blvm_srnn_latent_fwd(_pack_srnn(params), ptr(d), ptr(a), ptr(_f32c(z0)) if z0 is not None else None, ptr(eps),
                                     Tp, B, H, Z, R, int(residual), sd_eps, slope, ptr(zs), ptr(mu_q), ptr(sd_q), ptr(mu_p),
                                     ptr(sd_p), ptr(reserve), stream_ptr()),
            "blvm_srnn_latent_fwd",
        )  # fmt: skip
        _tick("fwd_end")
        kld = torch.zeros(B, device=d.device, dtype=torch.float64)
        kld_fn = torch.zeros(B, device=d.device, dtype=torch.float64)
        check(
            lib.blvm_kl_fwd(ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), LAYOUT_TIME_MAJOR, ptr(x_sl_dev), B, Tp, Z, stride,
                            fn_floor, ptr(kld), ptr(kld_fn), stream_ptr()),
            "blvm_kl_fwd",
        )  # fmt: skip
        ctx.cfg = cfg
        ctx.has_z0 = z0 is not None
        ctx.save_for_backward(d, a, eps, x_sl_dev, zs, mu_q, sd_q, mu_p, sd_p, reserve, *params)
        ctx.mark_non_differentiable(mu_q, sd_q, mu_p, sd_p)
        ctx.set_materialize_grads(False)  # (else autograd fills a [T',B,Z] zero gradient per statistics output: 16 MB each at [64,16000])
        return zs, kld, kld_fn, mu_q, sd_q, mu_p, sd_p

    @staticmethod
    def backward(ctx, d_zs, g_kld, g_kld_fn, *_unused):
        Tp, B, H, Z, R, residual, sd_eps, slope, stride, fn_floor = ctx.cfg
        d, a, eps, x_sl_dev, zs, mu_q, sd_q, mu_p, sd_p, reserve, *params = ctx.saved_tensors
        lib = load()
        f32 = dict(device=d.device, dtype=torch.float32)
        d_zs = _f32c(d_zs) if d_zs is not None else torch.zeros_like(zs)
        d_z = d_zs[1:]  # rows 1.. are the sampled latents; row 0 is z0 (its direct gradient is added below)
        c_raw = g_kld.to(torch.float32).contiguous() if g_kld is not None else None
        c_fn = g_kld_fn.to(torch.float32).contiguous() if g_kld_fn is not None else None
        grads = _zeros_like_many(params)
        d_d, d_a = torch.empty_like(d), torch.empty_like(a)
        d_z0 = torch.empty(B, Z, **f32) if ctx.has_z0 else None
        ws = torch.empty(lib.blvm_srnn_bwd_workspace_floats(Tp, B, H, Z, R), **f32)
        _tick("bwd_begin")
        check(
            lib.blvm_srnn_latent_bwd(_pack_srnn(params), ptr(d), ptr(a), ptr(eps), ptr(zs), ptr(mu_q), ptr(sd_q), ptr(mu_p),
                                     ptr(sd_p), ptr(reserve), ptr(d_z), ptr(x_sl_dev), ptr(c_raw), ptr(c_fn), stride, fn_floor,
                                     Tp, B, H, Z, R, int(residual), sd_eps, slope, ptr(d_d), ptr(d_a), ptr(d_z0),
                                     _pack_srnn(grads), ptr(ws), stream_ptr()),
            "blvm_srnn_latent_bwd",
        )  # fmt: skip
        _tick("bwd_end")
        if d_z0 is not None:
            d_z0 = d_z0 + d_zs[0]
        return (d_d, d_a, d_z0, None, None, None, *grads)


def srnn_latent_chain(d, a, z0, eps, x_sl_dev, params, H, Z, R, residual_posterior, stride, free_nats=0.0, sd_eps=1e-6,
                      slope=LEAKY_SLOPE):  # fmt: skip
    """d, a [T',B,R] -> (zs [T'+1,B,Z] with zs[0]=z0, kld [B] f64, kld_fn [B] f64, mu_q, sd_q, mu_p, sd_p)."""
    Tp, B, _ = d.shape
    fn_floor = float(free_nats) / Z if free_nats else 0.0
    cfg = (Tp, B, H, Z, R, int(residual_posterior), float(sd_eps), float(slope), int(stride), fn_floor)  # 3: generate
    return _SRNNLatentFunction.apply(d, a, z0, eps, x_sl_dev, cfg, *params)


# ----------------------------------------------------------------------------------------------------------------------
# K10: WaveNet — dilated causal convolution (k=2) and the gated residual stack
# ----------------------------------------------------------------------------------------------------------------------


class _ScaleActFunction(torch.autograd.Function):
    """y = act(scale * x) (ReLU / LeakyReLU) element-wise."""

    @staticmethod
    def forward(ctx, x, scale, slope):
        x = _f32c(x)
        y = torch.empty_like(x)
        check(load().blvm_scale_act_f32(ptr(x), scale, slope, ptr(y), x.numel(), stream_ptr()), "blvm_scale_act_f32")
        ctx.save_for_backward(y)
        ctx.cfg = (scale, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        scale, slope = ctx.cfg
        dy = _f32c(dy)
        dz = torch.empty_like(dy)
        check(load().blvm_act_bwd_f32(ptr(dy), ptr(y), slope, ptr(dz), dz.numel(), stream_ptr()), "blvm_act_bwd_f32")
        if scale != 1.0:  # chain rule through the scale: one more streaming pass over dz
            dx = torch.empty_like(dz)
            check(load().blvm_scale_act_f32(ptr(dz), scale, 1.0, ptr(dx), dz.numel(), stream_ptr()), "blvm_scale_act_f32")
            dz = dx
        return dz, None, None


def scale_act(x, scale: float = 1.0, slope: float = 0.0):
    return _ScaleActFunction.apply(x, float(scale), float(slope))


class _Conv1dK2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, dilation):
        x, W, b = _f32c(x), _f32c(W), _f32c(b)
        L, B, Cin = x.shape
        Cout = W.shape[0]
        lib = load()
        f32 = dict(device=x.device, dtype=torch.float32)
        out = torch.empty(L - dilation, B, Cout, **f32)
        ws = torch.empty(lib.blvm_conv1d_k2_workspace_floats(Cin, Cout), **f32)
        check(lib.blvm_conv1d_k2_fwd(ptr(x), ptr(W), ptr(b), L, B, Cin, Cout, dilation, ptr(out), ptr(ws), stream_ptr()),
              "blvm_conv1d_k2_fwd")  # fmt: skip
        ctx.save_for_backward(x, W)
        ctx.dims = (L, B, Cin, Cout, dilation)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, W = ctx.saved_tensors
        L, B, Cin, Cout, dilation = ctx.dims
        lib = load()
        f32 = dict(device=x.device, dtype=torch.float32)
        d_out = _f32c(d_out)
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW, db = torch.zeros_like(W), torch.zeros(Cout, **f32)
        ws = torch.empty(lib.blvm_conv1d_k2_workspace_floats(Cin, Cout), **f32)
        check(lib.blvm_conv1d_k2_bwd(ptr(x), ptr(W), ptr(d_out), L, B, Cin, Cout, dilation, ptr(d_x), ptr(dW), ptr(db), ptr(ws),
                                     stream_ptr()), "blvm_conv1d_k2_bwd")  # fmt: skip
        return d_x, dW, db, None


def conv1d_k2(x, weight, bias, dilation: int = 1):
    """Time-major dilated convolution with kernel size 2: x [L,B,Cin], weight [Cout,Cin,2] -> [L-dilation,B,Cout]."""
    return _Conv1dK2Function.apply(x, weight, bias, int(dilation))


class _WaveNetStackFunction(torch.autograd.Function):
    """All gated residual blocks of a residual stack in one autograd node: x [L,B,C] -> G skip tensors [T_skip,B,S].
    `groups[i]` names the output that block i's skip branch is accumulated into; -1 = the block's skip is not used at all
    (its skip half of the 1x1 convolution is not computed; the weight rows get a zero gradient).
    params: (conv.weight [2C,C,2], conv.bias, conv1x1rs.weight [C+S,C,1], conv1x1rs.bias) per block."""

    @staticmethod
    def forward(ctx, x, dilations, groups, T_skip, inv_std, S, *params):
        import ctypes

        x = _f32c(x)
        params = tuple(_f32c(p) for p in params)
        L, B, C = x.shape
        lib = load()
        f32 = dict(device=x.device, dtype=torch.float32)
        n_out = max(groups) + 1
        skips = [torch.zeros(T_skip, B, S, **f32) for _ in range(n_out)]
        n = max(i for i in range(len(dilations)) if groups[i] >= 0) + 1  # blocks after the last used skip cannot influence any output
        dil = (ctypes.c_int * n)(*dilations[:n])
        grp = (ctypes.c_int * n)(*groups[:n])
        na, nr = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(lib.blvm_wavenet_stack_floats(L, B, C, dil, n, ctypes.byref(na), ctypes.byref(nr)), "blvm_wavenet_stack_floats")
        acts = torch.empty(max(na.value, 1), **f32)  # residual outputs of blocks 0 .. n-2, back to back
        reserve = torch.empty(nr.value, **f32)
        ws = torch.empty(lib.blvm_wavenet_block_workspace_floats(L, B, C, S, 1), **f32)
        pp = (ctypes.c_void_p * (4 * n))(*[p.data_ptr() for p in params[: 4 * n]])
        sk = (ctypes.c_void_p * n_out)(*[t.data_ptr() for t in skips])
        check(lib.blvm_wavenet_stack_fwd(ptr(x), pp, dil, grp, n, L, B, C, S, T_skip, inv_std, ptr(acts), sk, ptr(reserve), ptr(ws),
                                         stream_ptr()), "blvm_wavenet_stack_fwd")  # fmt: skip
        ctx.cfg = (tuple(dilations), tuple(groups), T_skip, inv_std, S, L, B, C, n)
        ctx.save_for_backward(x, acts, reserve, *params)
        return tuple(skips)

    @staticmethod
    def backward(ctx, *d_skips):
        import ctypes

        dilations, groups, T_skip, inv_std, S, L, B, C, n = ctx.cfg
        x, acts, reserve, *params = ctx.saved_tensors
        lib = load()
        f32 = dict(device=x.device, dtype=torch.float32)
        d_skips = [_f32c(g) if g is not None else torch.zeros(T_skip, B, S, **f32) for g in d_skips]
        ws = torch.empty(lib.blvm_wavenet_block_workspace_floats(L, B, C, S, 1), **f32)
        grads = _zeros_like_many(params)
        d_x, d_scratch = torch.empty_like(x), torch.empty_like(x)
        dil = (ctypes.c_int * n)(*dilations[:n])
        grp = (ctypes.c_int * n)(*groups[:n])
        pp = (ctypes.c_void_p * (4 * n))(*[p.data_ptr() for p in params[: 4 * n]])
        gp = (ctypes.c_void_p * (4 * n))(*[g.data_ptr() for g in grads[: 4 * n]])
        ds = (ctypes.c_void_p * len(d_skips))(*[t.data_ptr() for t in d_skips])
        check(lib.blvm_wavenet_stack_bwd(ptr(x), pp, dil, grp, n, L, B, C, S, T_skip, inv_std, ptr(acts), ptr(reserve), ds, ptr(d_x),
                                         ptr(d_scratch), gp, ptr(ws), stream_ptr()), "blvm_wavenet_stack_bwd")  # fmt: skip
        return (d_x, None, None, None, None, None, *grads)


@torch.no_grad()
def wavenet_block_step(x2, params, inv_std: float, S: int, skip_acc, want_output: bool = True):
    """One gated residual block on ONE new frame for cached generation: x2 [2,B,C] = (the block's input `dilation` frames ago,
    its input now) -> the block's output now [1,B,C]; the skip branch is accumulated into skip_acc [1,B,S].  The same fused
    block kernel as training (K10) at L_in = 2, dilation = 1."""
    x2 = _f32c(x2)
    _, B, C = x2.shape
    cw, cb, rw, rb = (_f32c(p) for p in params)
    lib = load()
    f32 = dict(device=x2.device, dtype=torch.float32)
    o = torch.empty(1, B, C, **f32) if want_output else None
    res = torch.empty(lib.blvm_wavenet_block_reserve_floats(2, B, C, 1), **f32)
    ws = torch.empty(lib.blvm_wavenet_block_workspace_floats(2, B, C, S, 1), **f32)
    check(lib.blvm_wavenet_block_fwd(ptr(x2), ptr(cw), ptr(cb), ptr(rw), ptr(rb), 2, B, C, S, 1, 1, inv_std, ptr(o), ptr(skip_acc),
                                     ptr(res), ptr(ws), stream_ptr()), "blvm_wavenet_block_fwd")  # fmt: skip
    return o


@torch.no_grad()
def wavenet_decode(causal, in_transform, blocks_params, dilations, out_linear, head_linear, B: int, n_frames: int, inv_std: float,
                   skip_scale: float, num_mix: int, log_eps: float, u=None, v=None):
    """K10c: all frames of B utterances in one launch.  causal / in_transform / out_linear / head_linear = (weight, bias);
    blocks_params as for wavenet_stack; u [n_frames,B,num_mix], v [n_frames,B] uniform draws (None: the mode).  -> x [B,n_frames]."""
    import ctypes

    lib = load()
    C, S, O = in_transform[0].shape[0], blocks_params[0][2].shape[0] - in_transform[0].shape[0], out_linear[0].shape[0]
    dev = causal[0].device
    if causal[0].numel() != 2 * C:
        raise NotImplementedError("libblvm_hip: wavenet_decode is built for in_channels = n_stack_frames = 1")
    hw, hb = head_linear
    if hw.shape[0] > 32 or hw.shape[0] != 3 * num_mix:
        raise NotImplementedError("libblvm_hip: wavenet_decode head must have 3 * num_mix <= 32 outputs")
    pad_w = torch.zeros(32 - hw.shape[0], hw.shape[1], device=dev)
    parts = [*causal, *in_transform, *(p for blk in blocks_params for p in blk), *out_linear, hw, pad_w, hb, pad_w.new_zeros(32 - hb.numel())]
    packed = torch.cat([_f32c(p).reshape(-1) for p in parts])
    if packed.numel() != lib.blvm_wavenet_decode_pack_floats(C, S, O, len(blocks_params)):
        raise ValueError("wavenet_decode: parameter shapes do not match the packed layout")
    dil = (ctypes.c_int * len(dilations))(*dilations)
    queues = torch.empty(lib.blvm_wavenet_decode_scratch_floats(dil, len(dilations), B, C, S), device=dev, dtype=torch.float32)
    x = torch.empty(B, n_frames, device=dev, dtype=torch.float32)
    if u is not None:
        u, v = _f32c(u), _f32c(v)
        if tuple(u.shape) != (n_frames, B, num_mix) or tuple(v.shape) != (n_frames, B):
            raise ValueError("wavenet_decode: u must be [n_frames,B,num_mix] and v [n_frames,B]")
    check(lib.blvm_wavenet_decode(ptr(packed), dil, len(dilations), B, C, S, O, num_mix, n_frames, inv_std, skip_scale, log_eps,
                                  ptr(u), ptr(v), ptr(queues), ptr(x), stream_ptr()), "blvm_wavenet_decode")  # fmt: skip
    return x


def wavenet_stack(x, blocks_params, dilations, T_skip: int, inv_std: float, S: int, groups=None):
    """x [L,B,C] -> skip output(s) [T_skip,B,S]: the sum over blocks of the last T_skip frames of each block's skip branch
    (groups=None: one sum over all blocks, returned as a tensor; otherwise a tuple, see _WaveNetStackFunction)."""
    flat = [p for blk in blocks_params for p in blk]
    g = tuple(0 for _ in dilations) if groups is None else tuple(int(v) for v in groups)
    out = _WaveNetStackFunction.apply(x, tuple(int(d) for d in dilations), g, int(T_skip), float(inv_std), int(S), *flat)
    return out[0] if groups is None else out


# ----------------------------------------------------------------------------------------------------------------------
# K5: RSSM cell (Clockwork-VAE) over a sequence
# ----------------------------------------------------------------------------------------------------------------------

_RSSM_PARAM_ORDER = (
    ["gin_w", "gin_b", "gru_wih", "gru_whh", "gru_bih", "gru_bhh"]
    + ["prior_w0", "prior_b0", "prior_w1", "prior_b1", "prior_w2", "prior_b2", "prior_hw", "prior_hb"]
    + ["post_w0", "post_b0", "post_w1", "post_b1", "post_w2", "post_b2", "post_hw", "post_hb"]
)
RSSM_PLAIN, RSSM_RESIDUAL, RSSM_PRECISION, RSSM_GENERATE = 0, 1, 2, 3  # 3: z drawn from the prior (generation)


def _pack_rssm(ts):
    d = dict(zip(_RSSM_PARAM_ORDER, ts))
    w = _hip.RssmWeights()
    for k in ("gin_w", "gin_b", "gru_wih", "gru_whh", "gru_bih", "gru_bhh", "prior_hw", "prior_hb", "post_hw", "post_hb"):
        setattr(w, k, ptr(d[k]))
    for i in range(3):
        w.prior_w[i], w.prior_b[i] = ptr(d[f"prior_w{i}"]), ptr(d[f"prior_b{i}"])
        w.post_w[i], w.post_b[i] = ptr(d[f"post_w{i}"]), ptr(d[f"post_b{i}"])
    return w


class _RSSMSeqFunction(torch.autograd.Function):
    """(enc, ctx, z0, h0, eps, 22 params) -> zs [T+1,B,Z], hs [T+1,B,H], kld [B], kld_fn [B] (+ non-diff. mu/sd)."""

    @staticmethod
    def forward(ctx_, enc, ctx, z0, h0, eps, x_sl_dev, cfg, *params):
        T, B, H, Z, C, E, mode, sd_eps, stride, fn_floor = cfg
        enc, eps = _f32c(enc), _f32c(eps)
        ctx = _f32c(ctx) if ctx is not None else None
        params = tuple(_f32c(p) for p in params)
        lib = load()
        f32 = dict(device=enc.device, dtype=torch.float32)
        zs, hs = torch.empty(T + 1, B, Z, **f32), torch.empty(T + 1, B, H, **f32)
        mu_q, sd_q, mu_p, sd_p = (torch.empty(T, B, Z, **f32) for _ in range(4))
        reserve = torch.empty(lib.blvm_rssm_reserve_floats(T, B, H, Z), **f32)
        check(
            lib.blvm_rssm_seq_fwd(_pack_rssm(params), ptr(enc), ptr(ctx), ptr(_f32c(z0)) if z0 is not None else None,
                                  ptr(_f32c(h0)) if h0 is not None else None, ptr(eps), T, B, H, Z, C, E, mode, sd_eps, ptr(zs),
                                  ptr(hs), ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), ptr(reserve), stream_ptr()),
            "blvm_rssm_seq_fwd",
        )  # fmt: skip
        kld = torch.zeros(B, device=enc.device, dtype=torch.float64)
        kld_fn = torch.zeros(B, device=enc.device, dtype=torch.float64)
        check(
            lib.blvm_kl_fwd(ptr(mu_q), ptr(sd_q), ptr(mu_p), ptr(sd_p), LAYOUT_TIME_MAJOR, ptr(x_sl_dev), B, T, Z, stride,
                            fn_floor, ptr(kld), ptr(kld_fn), stream_ptr()),
            "blvm_kl_fwd",
        )  # fmt: skip
        ctx_.cfg = cfg
        ctx_.has = (ctx is not None, z0 is not None, h0 is not None)
        saved = [enc, eps, x_sl_dev, zs, hs, mu_q, sd_q, mu_p, sd_p, reserve] + ([ctx] if ctx is not None else [])
        ctx_.n_fixed = len(saved)
        ctx_.save_for_backward(*saved, *params)
        ctx_.mark_non_differentiable(mu_q, sd_q, mu_p, sd_p)
        ctx_.set_materialize_grads(False)  # (else autograd fills a [T',B,Z] zero gradient per statistics output: 16 MB each at [64,16000])
        return zs, hs, kld, kld_fn, mu_q, sd_q, mu_p, sd_p

    @staticmethod
    def backward(ctx_, d_zs, d_hs, g_kld, g_kld_fn, *_unused):
        T, B, H, Z, C, E, mode, sd_eps, stride, fn_floor = ctx_.cfg
        has_ctx, has_z0, has_h0 = ctx_.has
        saved = ctx_.saved_tensors
        enc, eps, x_sl_dev, zs, hs, mu_q, sd_q, mu_p, sd_p, reserve = saved[:10]
        ctx = saved[10] if has_ctx else None
        params = saved[ctx_.n_fixed :]
        lib = load()
        f32 = dict(device=enc.device, dtype=torch.float32)
        d_zs = _f32c(d_zs) if d_zs is not None else torch.zeros_like(zs)
        d_hs = _f32c(d_hs) if d_hs is not None else torch.zeros_like(hs)
        c_raw = g_kld.to(torch.float32).contiguous() if g_kld is not None else None
        c_fn = g_kld_fn.to(torch.float32).contiguous() if g_kld_fn is not None else None
        grads = _zeros_like_many(params)
        d_enc = torch.empty_like(enc)
        d_ctx = torch.empty_like(ctx) if has_ctx else None
        d_z0 = torch.empty(B, Z, **f32) if has_z0 else None
        d_h0 = torch.empty(B, H, **f32) if has_h0 else None
        ws = torch.empty(lib.blvm_rssm_bwd_workspace_floats(T, B, H, Z), **f32)
        check(
            lib.blvm_rssm_seq_bwd(_pack_rssm(params), ptr(enc), ptr(ctx), ptr(eps), ptr(zs), ptr(hs), ptr(mu_q), ptr(sd_q),
                                  ptr(mu_p), ptr(sd_p), ptr(reserve), ptr(d_zs), ptr(d_hs), ptr(x_sl_dev), ptr(c_raw), ptr(c_fn),
                                  stride, fn_floor, T, B, H, Z, C, E, mode, sd_eps, ptr(d_enc), ptr(d_ctx), ptr(d_z0), ptr(d_h0),
                                  _pack_rssm(grads), ptr(ws), stream_ptr()),
            "blvm_rssm_seq_bwd",
        )  # fmt: skip
        if d_h0 is not None:
            d_h0 = d_h0 + d_hs[0]  # the direct gradient wrt the carried state (tiny [B,H] add)
        return (d_enc, d_ctx, d_z0, d_h0, None, None, None, *grads)


def rssm_sequence(enc, ctx, z0, h0, eps, x_sl_dev, params, H, Z, mode, stride, free_nats=0.0, sd_eps=1e-6):
    """enc [T,B,E], ctx [T,B,C] or None -> (zs [T+1,B,Z], hs [T+1,B,H], kld [B], kld_fn [B], mu_q, sd_q, mu_p, sd_p)."""
    T, B, E = enc.shape
    C = ctx.shape[2] if ctx is not None else 0
    fn_floor = float(free_nats) / Z if free_nats else 0.0
    cfg = (T, B, H, Z, C, E, int(mode), float(sd_eps), int(stride), fn_floor)
    return _RSSMSeqFunction.apply(enc, ctx, z0, h0, eps, x_sl_dev, cfg, *params)


# ----------------------------------------------------------------------------------------------------------------------
# K11: Clockwork-VAE convolutional coders — time-major channel-last [L,B,C]
# ----------------------------------------------------------------------------------------------------------------------


def _norm_ws(N, device):
    return torch.empty(load().blvm_chan_norm_workspace_doubles(N), device=device, dtype=torch.float64)


def _chan_norm_fwd(x, gamma, beta, eps):
    L, B, C = x.shape
    y, mr = torch.empty_like(x), torch.empty(2, B * C, device=x.device, dtype=torch.float32)
    check(load().blvm_chan_norm_fwd(ptr(x), L, B * C, C, ptr(gamma), ptr(beta), eps, ptr(y), ptr(mr), ptr(_norm_ws(B * C, x.device)),
                                    stream_ptr()), "blvm_chan_norm_fwd")  # fmt: skip
    return y, mr


def _chan_norm_stats(x, gamma, beta, eps):
    """Statistics only: mr [2,N] = [mean | rstd] and the column affine ss [2,N] with norm(x) = x * ss[0] + ss[1]."""
    L, B, C = x.shape
    mr, ss = (torch.empty(2, B * C, device=x.device, dtype=torch.float32) for _ in range(2))
    check(load().blvm_chan_norm_stats(ptr(x), L, B * C, C, ptr(gamma), ptr(beta), eps, ptr(mr), ptr(ss),
                                      ptr(_norm_ws(B * C, x.device)), stream_ptr()), "blvm_chan_norm_stats")  # fmt: skip
    return mr, ss


def _chan_norm_bwd(x, dy, mr, gamma, relu_mask, dgamma, dbeta, dx_chan_sum=None):
    L, B, C = x.shape
    dx = torch.empty_like(x)
    check(load().blvm_chan_norm_bwd(ptr(x), ptr(dy), ptr(mr), ptr(gamma), L, B * C, C, int(relu_mask), ptr(dx), ptr(dgamma),
                                    ptr(dbeta), ptr(dx_chan_sum), ptr(_norm_ws(B * C, x.device)), stream_ptr()),
          "blvm_chan_norm_bwd")  # fmt: skip
    return dx


def _dwconv_fwd(x, w, bias, stride, dilation, transposed, relu, in_affine=None):
    L, B, C = x.shape
    k = w.shape[-1]
    L_out = load().blvm_dwconv_out_length(L, k, stride, dilation, int(transposed))
    if L_out <= 0:
        raise _hip.BlvmHipError(f"depthwise conv: input length {L} is shorter than the kernel ({k=}, {dilation=})")
    y = torch.empty(L_out, B, C, device=x.device, dtype=torch.float32)
    sc, sh = (in_affine[0], in_affine[1]) if in_affine is not None else (None, None)
    check(load().blvm_dwconv_fwd(ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(bias), L, B * C, C, k, stride, dilation, int(transposed),
                                 int(relu), ptr(y), stream_ptr()), "blvm_dwconv_fwd")  # fmt: skip
    return y


def _dwconv_bwd(x, w, y, dy, stride, dilation, transposed, relu, need_dx, dw, dbias, in_affine=None):
    L, B, C = x.shape
    k = w.shape[-1]
    dx = torch.empty_like(x) if need_dx else None
    sc, sh = (in_affine[0], in_affine[1]) if in_affine is not None else (None, None)
    check(load().blvm_dwconv_bwd(ptr(x), ptr(sc), ptr(sh), ptr(w), ptr(y), ptr(dy), L, B * C, C, k, stride, dilation,
                                 int(transposed), int(relu), ptr(dx), ptr(dw), ptr(dbias), stream_ptr()), "blvm_dwconv_bwd")  # fmt: skip
    return dx


class _ChanNormFunction(torch.autograd.Function):
    """nn.GroupNorm(num_groups=C, C) on [L,B,C]: per-(sample, channel) normalisation over time."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        y, mr = _chan_norm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, mr, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mr, gamma = ctx.saved_tensors
        dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
        dx = _chan_norm_bwd(x, _f32c(dy), mr, gamma, False, dgamma, dbeta)
        return dx, dgamma, dbeta, None


def chan_norm(x, gamma, beta, eps: float = 1e-5):
    return _ChanNormFunction.apply(x, gamma, beta, eps)


class _DwConvFunction(torch.autograd.Function):
    """Depthwise Conv1d / ConvTranspose1d (groups = channels, no padding) on [L,B,C], optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, dilation, transposed, relu):
        x, w = _f32c(x), _f32c(w)
        bias = _f32c(bias) if bias is not None else None
        y = _dwconv_fwd(x, w, bias, stride, dilation, transposed, relu)
        ctx.cfg = (stride, dilation, transposed, relu, bias is not None)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, dilation, transposed, relu, has_bias = ctx.cfg
        x, w, y = ctx.saved_tensors
        dw = torch.zeros_like(w)
        db = torch.zeros(w.shape[0], device=w.device, dtype=torch.float32) if has_bias else None
        dx = _dwconv_bwd(x, w, y, _f32c(dy), stride, dilation, transposed, relu, ctx.needs_input_grad[0], dw, db)
        return dx, dw, db, None, None, None, None


def dwconv(x, weight, bias, stride=1, dilation=1, transposed=False, relu=False):
    return _DwConvFunction.apply(x, weight, bias, int(stride), int(dilation), bool(transposed), bool(relu))


class _ResampleAddFunction(torch.autograd.Function):
    """TemporalResidual: y + nearest-resampled x (convolutional_coders.py:15-26); y [Ly,B,C], x [Lx,B,C]."""

    @staticmethod
    def forward(ctx, y, x):
        y, x = _f32c(y), _f32c(x)
        Ly, B, C = y.shape
        out = torch.empty_like(y)
        check(load().blvm_resample_add_fwd(ptr(y), ptr(x), Ly, x.shape[0], B * C, ptr(out), stream_ptr()), "blvm_resample_add_fwd")
        ctx.Lx = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32c(dout)
        Ly, B, C = dout.shape
        if ctx.Lx == Ly:
            return dout, dout
        dx = torch.zeros(ctx.Lx, B, C, device=dout.device, dtype=torch.float32)
        check(load().blvm_resample_add_bwd(ptr(dout), Ly, ctx.Lx, B * C, ptr(dx), stream_ptr()), "blvm_resample_add_bwd")
        return dout, dx


def resample_add(y, x):
    return _ResampleAddFunction.apply(y, x)


def dense_conv(x, weight, bias, stride: int = 1, dilation: int = 1, transposed: bool = False):
    """Dense k-tap Conv1d / ConvTranspose1d (groups = 1, no padding) on time-major x [L,B,Cin] -> [L_out,B,Cout]: the products of
    ALL taps are one K6 GEMM over every input row ([L*B, Cin] x [Cin, k*Cout], its weight and input gradients from the same node);
    the taps are then strided slices of that product, summed (convolution) or added into their strided output positions
    (transposed convolution) by autograd-tracked slicing.  weight: Conv1d [Cout,Cin,k], ConvTranspose1d [Cin,Cout,k].
    (`BlockSimple`, convolutional_coders.py:69-91 — not a BASELINE configuration: the stride-s convolution multiplies s times the
    rows it keeps.)"""
    L, B, Cin = x.shape
    k = weight.shape[-1]
    Cout = weight.shape[1] if transposed else weight.shape[0]
    wcat = (weight.permute(2, 1, 0) if transposed else weight.permute(2, 0, 1)).reshape(k * Cout, Cin)  # row j * Cout + co
    y = linear(x.reshape(L * B, Cin), wcat, None).view(L, B, k, Cout)
    if transposed:
        L_out = (L - 1) * stride + dilation * (k - 1) + 1
        out = x.new_zeros(L_out, B, Cout)
        for j in range(k):
            sl = slice(j * dilation, j * dilation + (L - 1) * stride + 1, stride)
            out[sl] = out[sl] + y[:, :, j]
    else:
        L_out = (L - dilation * (k - 1) - 1) // stride + 1
        if L_out < 1:
            raise ValueError(f"dense_conv: input of {L} frames is shorter than the kernel's extent {dilation * (k - 1) + 1}")
        out = sum(y[j * dilation : j * dilation + (L_out - 1) * stride + 1 : stride, :, j] for j in range(k))
    return out + bias if bias is not None else out


class _SepBlockFunction(torch.autograd.Function):
    """One `BlockSeparable` (convolutional_coders.py:29-66) as a single autograd node on [L,B,C]:
    1x1 conv C->4C + ReLU (K6 epilogue) -> channel norm -> depthwise (transposed) conv k, stride s + ReLU -> channel norm
    -> 1x1 conv 4C->C (no bias) -> + nearest-resampled input.  params: w1 [4C,C], b1, g1, be1, wd [4C,k], bd, g2, be2,
    wp [C,4C]."""

    @staticmethod
    def forward(ctx, x, cfg, w1, b1, g1, be1, wd, bd, g2, be2, wp):
        stride, dilation, transposed, eps = cfg
        x = _f32c(x)
        w1, b1, g1, be1, wd, bd, g2, be2, wp = (_f32c(p) for p in (w1, b1, g1, be1, wd, bd, g2, be2, wp))
        L, B, C = x.shape
        Cb = w1.shape[0]
        a1 = torch.empty(L, B, Cb, device=x.device, dtype=torch.float32)
        gemm(0, 0, L * B, Cb, C, x, C, w1, C, a1, Cb, bias=b1, act=ACT_RELU)
        mr1, ss1 = _chan_norm_stats(a1, g1, be1, eps)  # norm 1 is applied inside the stencil's loads: never materialised
        d = _dwconv_fwd(a1, wd, bd, stride, dilation, transposed, True, in_affine=ss1)
        n2, mr2 = _chan_norm_fwd(d, g2, be2, eps)
        L2 = d.shape[0]
        r = torch.empty(L2, B, C, device=x.device, dtype=torch.float32)
        gemm(0, 0, L2 * B, C, Cb, n2, Cb, wp, Cb, r, C)
        out = torch.empty_like(r)
        check(load().blvm_resample_add_fwd(ptr(r), ptr(x), L2, L, B * C, ptr(out), stream_ptr()), "blvm_resample_add_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(x, a1, mr1, ss1, d, mr2, n2, w1, g1, wd, g2, wp)
        return out

    @staticmethod
    def backward(ctx, dout):
        stride, dilation, transposed, eps = ctx.cfg
        x, a1, mr1, ss1, d, mr2, n2, w1, g1, wd, g2, wp = ctx.saved_tensors
        dout = _f32c(dout)
        L, B, C = x.shape
        L2, Cb = d.shape[0], w1.shape[0]
        M, M2 = L * B, L2 * B
        dev = x.device
        # pointwise 4C->C
        dwp, dg2, dbe2, dwd, dbd, dg1, dbe1, dw1, db1 = _zeros_like_many([wp, g2, g2, wd, g2, g1, g1, w1, g1])
        gemm(1, 1, C, Cb, M2, dout, C, n2, Cb, dwp, Cb, accumulate=True, split_k=_pick_split(C, Cb, M2))
        dn2 = torch.empty(L2, B, Cb, device=dev, dtype=torch.float32)
        gemm(0, 1, M2, Cb, C, dout, C, wp, Cb, dn2, Cb)
        # norm 2 (+ the ReLU in front of it: d > 0 <=> pre-activation > 0)
        dd = _chan_norm_bwd(d, dn2, mr2, g2, True, dg2, dbe2)
        del dn2
        # depthwise conv
        dn1 = _dwconv_bwd(a1, wd, None, dd, stride, dilation, transposed, False, True, dwd, dbd, in_affine=ss1)
        del dd
        # norm 1 (+ ReLU of the 1x1 conv)
        da1 = _chan_norm_bwd(a1, dn1, mr1, g1, True, dg1, dbe1, dx_chan_sum=db1)  # db1 = column sums of da1, same pass
        del dn1
        # 1x1 conv C->4C
        gemm(1, 1, Cb, C, M, da1, Cb, x, C, dw1, C, accumulate=True, split_k=_pick_split(Cb, C, M))
        dx = None
        if ctx.needs_input_grad[0]:
            if L2 == L:
                dx = dout.clone()
            else:
                dx = torch.zeros_like(x)
                check(load().blvm_resample_add_bwd(ptr(dout), L2, L, B * C, ptr(dx), stream_ptr()), "blvm_resample_add_bwd")
            gemm(0, 1, M, C, Cb, da1, Cb, w1, C, dx, C, accumulate=True)
        return dx, None, dw1, db1, dg1, dbe1, dwd, dbd, dg2, dbe2, dwp


def sep_block(x, stride, dilation, transposed, w1, b1, g1, be1, wd, bd, g2, be2, wp, eps: float = 1e-5):
    return _SepBlockFunction.apply(x, (int(stride), int(dilation), bool(transposed), float(eps)), w1, b1, g1, be1, wd, bd, g2, be2, wp)
