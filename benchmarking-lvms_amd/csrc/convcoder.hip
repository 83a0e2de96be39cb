// convcoder.hip — K11: the streaming pieces of the Clockwork-VAE's depthwise-separable convolutional coders.
//
// Replaces, inside `BlockSeparable` (blvm/models/clockwork_vae/convolutional_coders.py:29-66) and
// `ConvDepthwiseSeparable1d` / `ConvTransposeDepthwiseSeparable1d` (blvm/modules/convolutions.py:6-104):
//   nn.GroupNorm(num_groups = num_channels)  -> per-(sample, channel) normalisation over TIME        (chan_norm)
//   nn.Conv1d / nn.ConvTranspose1d(groups = channels, kernel 5, stride s)                              (dwconv)
//   TemporalResidual's nearest-neighbour resampled skip (convolutional_coders.py:15-26)               (resample_add)
// The 1x1 convolutions of the block are dense [rows, C] GEMMs on K6.  Layout: TIME-MAJOR channel-last [L, B, C]
// viewed as a row-major matrix [L, N = B*C]: a (sample, channel) pair is a COLUMN, time runs down the rows, so the
// normalisation statistics are column reductions and the depthwise stencils touch the same columns of a few rows —
// every access is a coalesced 16-byte-per-lane stream.  All kernels are HBM-bound.
#include "common.h"

namespace blvm {
namespace {

inline dim3 ew_grid(size_t n_items) {
  size_t blocks = (n_items + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// Thread geometry of every streaming kernel below: a 256-thread workgroup covers 64 COLUMN GROUPS (4 adjacent columns =
// one 16-byte access per lane, 1 KiB contiguous per wave and row) x 4 row strips (one per wave); per-column constants
// (weights, statistics, affine) are loaded once per thread and kept in registers while it walks down its strip.
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 relu_mask4(float4 v, float4 m) {
  return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
}

inline int pick_rows_per_block(int L, int col_blocks, int min_rows) {
  // enough workgroups to fill 256 CUs a few times over, strips long enough to amortise the per-thread constants
  int want_blocks = (4096 + col_blocks - 1) / col_blocks;
  int rpb = (L + want_blocks - 1) / want_blocks;
  if (rpb < min_rows) rpb = min_rows;
  return (rpb + 3) / 4 * 4;
}

// ---- column statistics in float64 ------------------------------------------------------------------------------------
// mode 0: s1 = sum_t x, s2 = sum_t x^2           (forward statistics)
// mode 1: s1 = sum_t dy, s2 = sum_t dy * xhat    (backward), xhat = (x - mean) * rstd with mr = [mean | rstd]
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ mr, int L, int N, int rows_per_block,
                                                        int mode, double* __restrict__ out) {
  __shared__ double part[4][8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (n < N) {
    if (mode == 0) {
#pragma unroll 4
      for (int r = r0 + wave; r < r1; r += 4) {
        const float4 v = ld4g(x + (size_t)r * N + n);
        s1[0] += (double)v.x; s2[0] += (double)v.x * (double)v.x;
        s1[1] += (double)v.y; s2[1] += (double)v.y * (double)v.y;
        s1[2] += (double)v.z; s2[2] += (double)v.z * (double)v.z;
        s1[3] += (double)v.w; s2[3] += (double)v.w * (double)v.w;
      }
    } else {
      const float4 mean = ld4g(mr + n), rstd = ld4g(mr + N + n);
#pragma unroll 4
      for (int r = r0 + wave; r < r1; r += 4) {
        const float4 g = ld4g(dy + (size_t)r * N + n);
        const float4 v = ld4g(x + (size_t)r * N + n);
        s1[0] += (double)g.x; s2[0] += (double)g.x * (double)((v.x - mean.x) * rstd.x);
        s1[1] += (double)g.y; s2[1] += (double)g.y * (double)((v.y - mean.y) * rstd.y);
        s1[2] += (double)g.z; s2[2] += (double)g.z * (double)((v.z - mean.z) * rstd.z);
        s1[3] += (double)g.w; s2[3] += (double)g.w * (double)((v.w - mean.w) * rstd.w);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { part[wave][q][lane] = s1[q]; part[wave][4 + q][lane] = s2[q]; }
  __syncthreads();
  if (wave == 0 && n < N) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      atomicAdd(out + n + q, part[0][q][lane] + part[1][q][lane] + part[2][q][lane] + part[3][q][lane]);
      atomicAdd(out + N + n + q, part[0][4 + q][lane] + part[1][4 + q][lane] + part[2][4 + q][lane] + part[3][4 + q][lane]);
    }
  }
}

// sums -> mr = [mean | rstd] and the column affine ss = [scale | shift] with norm(x) = x*scale + shift
__global__ __launch_bounds__(256) void finalize_stats_kernel(const double* __restrict__ sums, int L, int N, int C, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mr, float* __restrict__ ss) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double mean = sums[n] / L;
  double var = sums[N + n] / L - mean * mean;  // biased variance, as nn.GroupNorm
  if (var < 0.0) var = 0.0;
  const float m = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
  mr[n] = m;
  mr[N + n] = rs;
  if (ss != nullptr) {
    const float sc = rs * gamma[n % C];
    ss[n] = sc;
    ss[N + n] = beta[n % C] - m * sc;  // (x - mean) * rstd * gamma + beta == x * sc + (beta - mean * sc)
  }
}

// y = (x - mean) * rstd * gamma_c + beta_c, evaluated exactly in that order (bit-identical to the unfused form)
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int L,
                                                         int N, int C, int rows_per_block, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  const float4 m = ld4g(mr + n), s = ld4g(mr + N + n), g = ld4g(gamma + n % C), b = ld4g(beta + n % C);
#pragma unroll 4
  for (int r = r0 + wave; r < r1; r += 4) {
    const float4 v = ld4g(x + (size_t)r * N + n);
    float4 o;
    o.x = (v.x - m.x) * s.x * g.x + b.x; o.y = (v.y - m.y) * s.y * g.y + b.y;
    o.z = (v.z - m.z) * s.z * g.z + b.z; o.w = (v.w - m.w) * s.w * g.w + b.w;
    st4g(y + (size_t)r * N + n, o);
  }
}

// dx = gamma_c * rstd * (dy - s1/L - xhat * s2/L)   [optionally * (x > 0): ReLU in front of the norm]
// dsum (optional, [C]): += sum over rows and samples of dx — the bias gradient of the layer that produced x.
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ mr, const double* __restrict__ sums,
                                                             const float* __restrict__ gamma, int L, int N, int C,
                                                             int rows_per_block, int relu_mask, float* __restrict__ dx,
                                                             float* __restrict__ dsum) {
  __shared__ float part[4][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    const double invL = 1.0 / (double)L;
    const float4 mean = ld4g(mr + n), rstd = ld4g(mr + N + n), g = ld4g(gamma + n % C);
    float m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { m1[q] = (float)(sums[n + q] * invL); m2[q] = (float)(sums[N + n + q] * invL); }
    const float gr[4] = {g.x * rstd.x, g.y * rstd.y, g.z * rstd.z, g.w * rstd.w};
#pragma unroll 4
    for (int r = r0 + wave; r < r1; r += 4) {
      const float4 xv = ld4g(x + (size_t)r * N + n);
      const float4 gy = ld4g(dy + (size_t)r * N + n);
      float4 o;
      o.x = gr[0] * (gy.x - m1[0] - (xv.x - mean.x) * rstd.x * m2[0]);
      o.y = gr[1] * (gy.y - m1[1] - (xv.y - mean.y) * rstd.y * m2[1]);
      o.z = gr[2] * (gy.z - m1[2] - (xv.z - mean.z) * rstd.z * m2[2]);
      o.w = gr[3] * (gy.w - m1[3] - (xv.w - mean.w) * rstd.w * m2[3]);
      if (relu_mask) o = relu_mask4(o, xv);
      st4g(dx + (size_t)r * N + n, o);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
  }
  if (dsum == nullptr) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) part[wave][q][lane] = acc[q];
  __syncthreads();
  if (wave == 0 && n < N) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      atomicAdd(dsum + (n + q) % C, part[0][q][lane] + part[1][q][lane] + part[2][q][lane] + part[3][q][lane]);
  }
}

// dgamma_c += sum_b s2[b,c];  dbeta_c += sum_b s1[b,c]
__global__ __launch_bounds__(256) void norm_param_grad_kernel(const double* __restrict__ sums, int B, int C, int N,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double g = 0.0, b = 0.0;
  for (int i = 0; i < B; ++i) { b += sums[i * C + c]; g += sums[N + i * C + c]; }
  if (dgamma) dgamma[c] += (float)g;
  if (dbeta) dbeta[c] += (float)b;
}

// ---- depthwise stencils ------------------------------------------------------------------------------------------------
// gather form:  y[u] = bias + sum_j w_j * x[u*s + j*d]            (Conv1d forward; ConvTranspose1d backward-data)
// scatter form: y[t] = bias + sum_j [ (t - j*d) % s == 0 ] w_j * x[(t - j*d)/s]   (ConvTranspose1d fwd; Conv1d bwd-data)
// Optional on the INPUT of the stencil, applied to in-range rows only:
//   `mask` (shape of x): x *= (mask > 0) — the ReLU derivative of the stencil's output when x is an incoming gradient;
//   `sc`/`sh` ([N] each): x = x*sc + sh — the per-(sample, channel) normalisation in front of the convolution, so the
//   normalised tensor never exists in memory.
// `relu`: apply ReLU to y.
struct DwArgs {
  const float* x;
  const float* mask;
  const float* w;     // [C,k]
  const float* bias;  // [C] or null
  const float* sc;
  const float* sh;
  float* y;
  int Lx, Ly, N, C, k, s, d, relu;
};

// generic fallback (any k <= 8, stride, dilation): one output element group per thread
template <bool SCATTER>
__global__ __launch_bounds__(256) void dw_stencil_kernel(DwArgs a) {
  const int n4 = a.N / 4;
  const size_t total = (size_t)a.Ly * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int u = (int)(i / n4);
    const int n = (int)(i - (size_t)u * n4) * 4, c = n % a.C;
    float4 acc = a.bias ? ld4g(a.bias + c) : zero4();
    const float4 sc = a.sc ? ld4g(a.sc + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = a.sc ? ld4g(a.sh + n) : zero4();
    for (int j = 0; j < a.k; ++j) {
      int r;
      if (!SCATTER) {
        r = u * a.s + j * a.d;
        if (r >= a.Lx) continue;
      } else {
        const int q = u - j * a.d;
        if (q < 0 || q % a.s != 0) continue;
        r = q / a.s;
        if (r >= a.Lx) continue;
      }
      float4 v = ld4g(a.x + (size_t)r * a.N + n);
      if (a.mask) v = relu_mask4(v, ld4g(a.mask + (size_t)r * a.N + n));
      if (a.sc) v = fma4(v, sc, sh);
      const float4 wj = make_float4(a.w[(c + 0) * a.k + j], a.w[(c + 1) * a.k + j], a.w[(c + 2) * a.k + j], a.w[(c + 3) * a.k + j]);
      acc = fma4(wj, v, acc);
    }
    if (a.relu) acc = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
    st4g(a.y + (size_t)u * a.N + n, acc);
  }
}

__device__ __forceinline__ float4 dw_load_row(const DwArgs& a, int r, int n, float4 sc, float4 sh) {
  if (r < 0 || r >= a.Lx) return zero4();
  float4 v = ld4g(a.x + (size_t)r * a.N + n);
  if (a.mask) v = relu_mask4(v, ld4g(a.mask + (size_t)r * a.N + n));
  if (a.sc) v = fma4(v, sc, sh);
  return v;
}

// K taps, stride S, dilation 1: each thread walks R consecutive outputs of its 4 columns with the K input rows of the
// current output in registers; advancing one output shifts the window by S rows, so every input row is LOADED ONCE per
// strip (+ K-S halo rows per strip) instead of once per tap.
template <int K, int S, int R>
__global__ __launch_bounds__(256) void dw_gather_strip_kernel(DwArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int u0 = (blockIdx.y * 4 + wave) * R;
  if (n >= a.N || u0 >= a.Ly) return;
  const int c = n % a.C;
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = make_float4(a.w[(c + 0) * K + j], a.w[(c + 1) * K + j], a.w[(c + 2) * K + j], a.w[(c + 3) * K + j]);
  const float4 bias = a.bias ? ld4g(a.bias + c) : zero4();
  const float4 sc = a.sc ? ld4g(a.sc + n) : zero4(), sh = a.sc ? ld4g(a.sh + n) : zero4();
  float4 win[K];
#pragma unroll
  for (int j = 0; j < K; ++j) win[j] = dw_load_row(a, u0 * S + j, n, sc, sh);
  const int u1 = min(a.Ly, u0 + R);
  for (int u = u0; u < u1; ++u) {
    float4 nxt[S < K ? S : K];  // rows entering the window for output u+1 (issued before this output's arithmetic)
#pragma unroll
    for (int j = 0; j < (S < K ? S : K); ++j) nxt[j] = (u + 1 < u1) ? dw_load_row(a, (u + 1) * S + (K - (S < K ? S : K)) + j, n, sc, sh) : zero4();
    float4 acc = bias;
#pragma unroll
    for (int j = 0; j < K; ++j) acc = fma4(w[j], win[j], acc);
    if (a.relu) acc = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
    st4g(a.y + (size_t)u * a.N + n, acc);
    if (S < K) {
#pragma unroll
      for (int j = 0; j + S < K; ++j) win[j] = win[j + S];
#pragma unroll
      for (int j = 0; j < S; ++j) win[K - S + j] = nxt[j];
    } else {
#pragma unroll
      for (int j = 0; j < K; ++j) win[j] = nxt[j];
    }
  }
}

// Scatter form by output phase: t = S*m + p takes the taps j = p + i*S of the inputs x[m - i]; a thread walks R input
// rows m with the ceil(K/S) most recent ones in registers and emits S output rows per input row.
template <int K, int S, int R>
__global__ __launch_bounds__(256) void dw_scatter_strip_kernel(DwArgs a) {
  constexpr int W = (K + S - 1) / S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int m0 = (blockIdx.y * 4 + wave) * R;
  if (n >= a.N || m0 * S >= a.Ly) return;
  const int c = n % a.C;
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = make_float4(a.w[(c + 0) * K + j], a.w[(c + 1) * K + j], a.w[(c + 2) * K + j], a.w[(c + 3) * K + j]);
  const float4 bias = a.bias ? ld4g(a.bias + c) : zero4();
  const float4 sc = a.sc ? ld4g(a.sc + n) : zero4(), sh = a.sc ? ld4g(a.sh + n) : zero4();
  float4 win[W];  // win[i] = x[m - i]
#pragma unroll
  for (int i = 1; i < W; ++i) win[i] = dw_load_row(a, m0 - i, n, sc, sh);
  win[0] = dw_load_row(a, m0, n, sc, sh);
  for (int m = m0; m < m0 + R && m * S < a.Ly; ++m) {
    const float4 nxt = dw_load_row(a, m + 1, n, sc, sh);
#pragma unroll
    for (int p = 0; p < S; ++p) {
      const int t = m * S + p;
      if (t >= a.Ly) break;
      float4 acc = bias;
#pragma unroll
      for (int i = 0; i < W; ++i)
        if (p + i * S < K) acc = fma4(w[p + i * S], win[i], acc);
      if (a.relu) acc = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
      st4g(a.y + (size_t)t * a.N + n, acc);
    }
#pragma unroll
    for (int i = W - 1; i > 0; --i) win[i] = win[i - 1];
    win[0] = nxt;
  }
}

// weight / bias gradient: dw[c,j] += sum_{u,b} A[u,b,c] * Bm[u*s + j*d, b, c], dbias[c] += sum A
// Conv1d: A = dout (mask = out), Bm = in.   ConvTranspose1d: A = in, Bm = dout (mask on Bm, bias from a separate sum).
// `sc`/`sh` normalise the convolution INPUT on load (A for the transposed form, Bm for the plain one).
struct DwWArgs {
  const float *A, *Bm, *mask, *sc, *sh;
  float *dw, *dbias;
  int LA, LB, N, C, k, s, d, rows_per_block, mask_on_b, bias_from_b, norm_on_a;
};

// generic fallback: one column per thread
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWArgs a) {
  __shared__ float part[4][64][9];  // up to 8 taps + bias
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int u0 = blockIdx.y * a.rows_per_block, u1 = min(a.LA, u0 + a.rows_per_block);
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.f;
  if (col < a.N) {
    const float sc = a.sc ? a.sc[col] : 1.f, sh = a.sc ? a.sh[col] : 0.f;
    for (int u = u0 + rl; u < u1; u += 4) {
      float av = a.A[(size_t)u * a.N + col];
      if (a.mask && !a.mask_on_b) av = a.mask[(size_t)u * a.N + col] > 0.f ? av : 0.f;
      if (a.sc && a.norm_on_a) av = fmaf(av, sc, sh);
      if (!a.bias_from_b) acc[8] += av;
      for (int j = 0; j < a.k; ++j) {
        const int r = u * a.s + j * a.d;
        if (r >= a.LB) break;
        float bv = a.Bm[(size_t)r * a.N + col];
        if (a.mask && a.mask_on_b) bv = a.mask[(size_t)r * a.N + col] > 0.f ? bv : 0.f;
        if (a.sc && !a.norm_on_a) bv = fmaf(bv, sc, sh);
        acc[j] = fmaf(av, bv, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) part[rl][threadIdx.x & 63][j] = acc[j];
  __syncthreads();
  if (rl == 0 && col < a.N) {
    const int i = threadIdx.x, c = col % a.C;
    for (int j = 0; j < a.k; ++j) atomicAdd(a.dw + c * a.k + j, part[0][i][j] + part[1][i][j] + part[2][i][j] + part[3][i][j]);
    if (a.dbias && !a.bias_from_b) atomicAdd(a.dbias + c, part[0][i][8] + part[1][i][8] + part[2][i][8] + part[3][i][8]);
  }
}

// K taps, stride S, dilation 1: 4 columns per thread, a contiguous strip of rows_per_block/4 rows of A per wave, the K
// rows of Bm under the current A row kept in registers (sliding window as in dw_gather_strip_kernel).
template <int K, int S>
__global__ __launch_bounds__(256) void dw_wgrad_strip_kernel(DwWArgs a) {
  __shared__ float part[4][(K + 1) * 4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int rows = a.rows_per_block / 4;
  const int u0 = blockIdx.y * a.rows_per_block + wave * rows, u1 = min(a.LA, u0 + rows);
  float4 acc[K + 1];
#pragma unroll
  for (int j = 0; j <= K; ++j) acc[j] = zero4();
  if (n < a.N && u0 < u1) {
    const float4 sc = a.sc ? ld4g(a.sc + n) : zero4(), sh = a.sc ? ld4g(a.sh + n) : zero4();
    // loads are unconditional (row index clamped, value selected afterwards): no branch sits between a load and the next
    // one, so the whole window refill of the following iteration is in flight under this iteration's FMAs
    auto ld_b = [&](int r) -> float4 {
      const bool ok = r < a.LB;
      const size_t off = (size_t)(ok ? r : a.LB - 1) * a.N + n;
      float4 v = ld4g(a.Bm + off);
      if (a.mask && a.mask_on_b) v = relu_mask4(v, ld4g(a.mask + off));
      if (a.sc && !a.norm_on_a) v = fma4(v, sc, sh);
      return ok ? v : zero4();
    };
    auto ld_a = [&](int u) -> float4 {
      const bool ok = u < u1;
      const size_t off = (size_t)(ok ? u : u1 - 1) * a.N + n;
      float4 v = ld4g(a.A + off);
      if (a.mask && !a.mask_on_b) v = relu_mask4(v, ld4g(a.mask + off));
      if (a.sc && a.norm_on_a) v = fma4(v, sc, sh);
      return ok ? v : zero4();
    };
    float4 win[K];
#pragma unroll
    for (int j = 0; j < K; ++j) win[j] = ld_b(u0 * S + j);
    float4 av = ld_a(u0);
    for (int u = u0; u < u1; ++u) {
      constexpr int NS = S < K ? S : K;
      float4 nxt[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) nxt[j] = ld_b((u + 1) * S + (K - NS) + j);
      const float4 av_next = ld_a(u + 1);
#pragma unroll
      for (int j = 0; j < K; ++j) acc[j] = fma4(av, win[j], acc[j]);
      acc[K].x += av.x; acc[K].y += av.y; acc[K].z += av.z; acc[K].w += av.w;
      if (S < K) {
#pragma unroll
        for (int j = 0; j + S < K; ++j) win[j] = win[j + S];
#pragma unroll
        for (int j = 0; j < NS; ++j) win[K - NS + j] = nxt[j];
      } else {
#pragma unroll
        for (int j = 0; j < K; ++j) win[j] = nxt[j];
      }
      av = av_next;
    }
  }
#pragma unroll
  for (int j = 0; j <= K; ++j) {
    part[wave][j * 4 + 0][lane] = acc[j].x; part[wave][j * 4 + 1][lane] = acc[j].y;
    part[wave][j * 4 + 2][lane] = acc[j].z; part[wave][j * 4 + 3][lane] = acc[j].w;
  }
  __syncthreads();
  // (K+1)*4 values x 64 lanes per workgroup: spread the adds over the 4 waves
  if (n < a.N) {
    for (int v = wave; v < (K + 1) * 4; v += 4) {
      const int j = v >> 2, q = v & 3, c = (n + q) % a.C;
      const float sum = part[0][v][lane] + part[1][v][lane] + part[2][v][lane] + part[3][v][lane];
      if (j < K) atomicAdd(a.dw + c * K + j, sum);
      else if (a.dbias && !a.bias_from_b) atomicAdd(a.dbias + c, sum);
    }
  }
}

// masked column sum over (rows, batch) per channel: dbias[c] += sum_{r,b} (mask > 0 ? x : 0)
__global__ __launch_bounds__(256) void masked_chan_sum_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                              int L, int N, int C, int rows_per_block, float* __restrict__ out) {
  __shared__ float part[4][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  float4 s = zero4();
  if (n < N) {
#pragma unroll 4
    for (int r = r0 + wave; r < r1; r += 4) {
      float4 v = ld4g(x + (size_t)r * N + n);
      if (mask != nullptr) v = relu_mask4(v, ld4g(mask + (size_t)r * N + n));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[wave][0][lane] = s.x; part[wave][1][lane] = s.y; part[wave][2][lane] = s.z; part[wave][3][lane] = s.w;
  __syncthreads();
  if (wave == 0 && n < N) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      atomicAdd(out + (n + q) % C, part[0][q][lane] + part[1][q][lane] + part[2][q][lane] + part[3][q][lane]);
  }
}

// ---- nearest-neighbour resampled residual ----------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = (int)floorf((float)dst * scale);  // torch 'nearest': floor(dst * (float)in/out), clamped
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ __launch_bounds__(256) void resample_add_kernel(const float* __restrict__ y, const float* __restrict__ x, int Ly,
                                                           int Lx, int N, float* __restrict__ out) {
  const int n4 = N / 4;
  const size_t total = (size_t)Ly * n4;
  const float scale = (float)Lx / (float)Ly;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / n4);
    const int n = (int)(i - (size_t)t * n4) * 4;
    const int r = Lx == Ly ? t : nearest_src(t, scale, Lx);
    const float4 a = *reinterpret_cast<const float4*>(y + (size_t)t * N + n);
    const float4 b = *reinterpret_cast<const float4*>(x + (size_t)r * N + n);
    *reinterpret_cast<float4*>(out + (size_t)t * N + n) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// dx[src(t)] += dout[t]  (several t may share a source row when up-sampling)
__global__ __launch_bounds__(256) void resample_add_bwd_kernel(const float* __restrict__ dout, int Ly, int Lx, int N,
                                                               float* __restrict__ dx) {
  const size_t total = (size_t)Ly * N;
  const float scale = (float)Lx / (float)Ly;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / N);
    const int n = (int)(i - (size_t)t * N);
    const int r = Lx == Ly ? t : nearest_src(t, scale, Lx);
    if (Ly <= Lx) dx[(size_t)r * N + n] += dout[i];  // injective map: no collisions
    else atomicAdd(dx + (size_t)r * N + n, dout[i]);
  }
}

int check_ln(int L, int N, int C) {
  BLVM_REQUIRE(L > 0 && N > 0 && C > 0 && N % C == 0 && C % 4 == 0, "convcoder: bad shape L=%d N=%d C=%d (C must be a multiple of 4)", L, N, C);
  return BLVM_OK;
}

int dw_len_out(int L_in, int k, int stride, int dilation, int transposed) {
  const int k_eff = dilation * (k - 1) + 1;
  return transposed ? (L_in - 1) * stride + k_eff : (L_in - k_eff) / stride + 1;
}

// gather / scatter stencil dispatch: register-window strip kernels for the coders' configurations (5 taps, stride 1/2/4,
// no dilation), the generic per-output kernel otherwise
template <bool SCATTER>
void launch_stencil(const DwArgs& a, hipStream_t s) {
  constexpr int R = 8;
  const int cb = (a.N / 4 + 63) / 64;
  if (a.k == 5 && a.d == 1 && (a.s == 1 || a.s == 2 || a.s == 4)) {
    if (!SCATTER) {
      dim3 grid(cb, (a.Ly + 4 * R - 1) / (4 * R));
      if (a.s == 1) hipLaunchKernelGGL((dw_gather_strip_kernel<5, 1, R>), grid, dim3(256), 0, s, a);
      else if (a.s == 2) hipLaunchKernelGGL((dw_gather_strip_kernel<5, 2, R>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((dw_gather_strip_kernel<5, 4, R>), grid, dim3(256), 0, s, a);
    } else {
      const int m_total = (a.Ly + a.s - 1) / a.s;
      dim3 grid(cb, (m_total + 4 * R - 1) / (4 * R));
      if (a.s == 1) hipLaunchKernelGGL((dw_scatter_strip_kernel<5, 1, R>), grid, dim3(256), 0, s, a);
      else if (a.s == 2) hipLaunchKernelGGL((dw_scatter_strip_kernel<5, 2, R>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((dw_scatter_strip_kernel<5, 4, R>), grid, dim3(256), 0, s, a);
    }
  } else {
    hipLaunchKernelGGL((dw_stencil_kernel<SCATTER>), ew_grid((size_t)a.Ly * (a.N / 4)), dim3(256), 0, s, a);
  }
}

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_chan_norm_workspace_doubles(int N) { return (size_t)2 * N; }

static int chan_norm_stats(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps, float* mr,
                           float* ss, double* workspace, hipStream_t s) {
  BLVM_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s));
  const int cb = (N / 4 + 63) / 64;
  const int rpb = pick_rows_per_block(L, cb, 64);
  hipLaunchKernelGGL(col_stats_kernel, dim3(cb, (L + rpb - 1) / rpb), dim3(256), 0, s, x, nullptr, nullptr, L, N, rpb, 0, workspace);
  hipLaunchKernelGGL(finalize_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, s, workspace, L, N, C, eps, gamma, beta, mr, ss);
  return BLVM_OK;
}

extern "C" int blvm_chan_norm_stats(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps,
                                    float* mr, float* scale_shift, double* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && gamma && beta && mr && scale_shift && workspace, "chan_norm_stats: null pointer");
  BLVM_REQUIRE(aligned16(x), "chan_norm_stats: alignment");
  rc = chan_norm_stats(x, L, N, C, gamma, beta, eps, mr, scale_shift, workspace, s);
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("chan_norm_stats");
  return BLVM_OK;
}

extern "C" int blvm_chan_norm_fwd(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps,
                                  float* y, float* mr, double* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && gamma && beta && y && mr && workspace, "chan_norm_fwd: null pointer");
  BLVM_REQUIRE(aligned16(x) && aligned16(y) && aligned16(mr) && aligned16(gamma) && aligned16(beta), "chan_norm_fwd: alignment");
  rc = chan_norm_stats(x, L, N, C, gamma, beta, eps, mr, nullptr, workspace, s);
  if (rc) return rc;
  const int cb = (N / 4 + 63) / 64;
  const int rpb = pick_rows_per_block(L, cb, 32);
  hipLaunchKernelGGL(norm_apply_kernel, dim3(cb, (L + rpb - 1) / rpb), dim3(256), 0, s, x, mr, gamma, beta, L, N, C, rpb, y);
  BLVM_CHECK_LAUNCH("chan_norm_fwd");
  return BLVM_OK;
}

extern "C" int blvm_chan_norm_bwd(const float* x, const float* dy, const float* mr, const float* gamma, int L, int N, int C,
                                  int relu_mask, float* dx, float* dgamma, float* dbeta, float* dx_chan_sum,
                                  double* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && dy && mr && gamma && dx && workspace, "chan_norm_bwd: null pointer");
  BLVM_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(mr) && aligned16(gamma), "chan_norm_bwd: alignment");
  BLVM_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s));
  const int cb = (N / 4 + 63) / 64;
  int rpb = pick_rows_per_block(L, cb, 64);
  hipLaunchKernelGGL(col_stats_kernel, dim3(cb, (L + rpb - 1) / rpb), dim3(256), 0, s, x, dy, mr, L, N, rpb, 1, workspace);
  rpb = pick_rows_per_block(L, cb, 32);
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(cb, (L + rpb - 1) / rpb), dim3(256), 0, s, x, dy, mr, workspace, gamma, L, N, C,
                     rpb, relu_mask, dx, dx_chan_sum);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(norm_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, N / C, C, N, dgamma, dbeta);
  BLVM_CHECK_LAUNCH("chan_norm_bwd");
  return BLVM_OK;
}

extern "C" int blvm_dwconv_out_length(int L_in, int k, int stride, int dilation, int transposed) {
  return dw_len_out(L_in, k, stride, dilation, transposed);
}

extern "C" int blvm_dwconv_fwd(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* bias,
                               int L_in, int N, int C, int k, int stride, int dilation, int transposed, int relu, float* y,
                               void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L_in, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && w && y && k > 0 && k <= 8 && stride > 0 && dilation > 0, "dwconv_fwd: bad arguments");
  BLVM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_fwd: in_scale and in_shift go together");
  const int L_out = dw_len_out(L_in, k, stride, dilation, transposed);
  BLVM_REQUIRE(L_out > 0, "dwconv_fwd: input of length %d is shorter than the kernel", L_in);
  BLVM_REQUIRE(aligned16(x) && aligned16(y) && (!in_scale || (aligned16(in_scale) && aligned16(in_shift))) && (!bias || aligned16(bias)),
               "dwconv_fwd: alignment");
  DwArgs a{x, nullptr, w, bias, in_scale, in_shift, y, L_in, L_out, N, C, k, stride, dilation, relu};
  if (transposed) launch_stencil<true>(a, s);
  else launch_stencil<false>(a, s);
  BLVM_CHECK_LAUNCH("dwconv_fwd");
  return BLVM_OK;
}

extern "C" int blvm_dwconv_bwd(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* y,
                               const float* dy, int L_in, int N, int C, int k, int stride, int dilation, int transposed, int relu,
                               float* dx, float* dw, float* dbias, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L_in, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && w && dy && k > 0 && k <= 8 && stride > 0 && dilation > 0, "dwconv_bwd: bad arguments");
  BLVM_REQUIRE(!relu || y, "dwconv_bwd: the ReLU mask needs the forward output");
  BLVM_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dwconv_bwd: in_scale and in_shift go together");
  BLVM_REQUIRE(aligned16(x) && aligned16(dy) && (!dx || aligned16(dx)) && (!y || aligned16(y)), "dwconv_bwd: alignment");
  const int L_out = dw_len_out(L_in, k, stride, dilation, transposed);
  const float* mask = relu ? y : nullptr;
  if (dx) {  // data gradient (wrt the affine-transformed input): the other stencil form applied to the (masked) output gradient
    DwArgs a{dy, mask, w, nullptr, nullptr, nullptr, dx, L_out, L_in, N, C, k, stride, dilation, 0};
    if (transposed) launch_stencil<false>(a, s);
    else launch_stencil<true>(a, s);
  }
  if (dw) {
    DwWArgs a{};
    a.dw = dw; a.dbias = dbias; a.N = N; a.C = C; a.k = k; a.s = stride; a.d = dilation; a.mask = mask;
    a.sc = in_scale; a.sh = in_shift;
    if (!transposed) { a.A = dy; a.Bm = x; a.LA = L_out; a.LB = L_in; a.mask_on_b = 0; a.bias_from_b = 0; a.norm_on_a = 0; }
    else { a.A = x; a.Bm = dy; a.LA = L_in; a.LB = L_out; a.mask_on_b = 1; a.bias_from_b = 1; a.norm_on_a = 1; }
    const int cb = (N / 4 + 63) / 64;
    if (k == 5 && dilation == 1 && (stride == 1 || stride == 2 || stride == 4)) {
      a.rows_per_block = pick_rows_per_block(a.LA, cb, 64);
      dim3 grid(cb, (a.LA + a.rows_per_block - 1) / a.rows_per_block);
      if (stride == 1) hipLaunchKernelGGL((dw_wgrad_strip_kernel<5, 1>), grid, dim3(256), 0, s, a);
      else if (stride == 2) hipLaunchKernelGGL((dw_wgrad_strip_kernel<5, 2>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((dw_wgrad_strip_kernel<5, 4>), grid, dim3(256), 0, s, a);
    } else {
      int rpb = 64;
      while ((a.LA + rpb - 1) / rpb > 4096) rpb *= 2;
      a.rows_per_block = rpb;
      hipLaunchKernelGGL(dw_wgrad_kernel, dim3((N + 63) / 64, (a.LA + rpb - 1) / rpb), dim3(256), 0, s, a);
    }
    if (transposed && dbias) {
      const int rb = pick_rows_per_block(L_out, cb, 64);
      hipLaunchKernelGGL(masked_chan_sum_kernel, dim3(cb, (L_out + rb - 1) / rb), dim3(256), 0, s, dy, mask, L_out, N, C, rb, dbias);
    }
  }
  BLVM_CHECK_LAUNCH("dwconv_bwd");
  return BLVM_OK;
}

extern "C" int blvm_resample_add_fwd(const float* y, const float* x, int L_out, int L_in, int N, float* out, void* stream_) {
  BLVM_REQUIRE(y && x && out && L_out > 0 && L_in > 0 && N > 0 && N % 4 == 0, "resample_add_fwd: bad arguments");
  hipLaunchKernelGGL(resample_add_kernel, ew_grid((size_t)L_out * (N / 4)), dim3(256), 0, static_cast<hipStream_t>(stream_), y, x,
                     L_out, L_in, N, out);
  BLVM_CHECK_LAUNCH("resample_add_fwd");
  return BLVM_OK;
}

extern "C" int blvm_resample_add_bwd(const float* dout, int L_out, int L_in, int N, float* dx, void* stream_) {
  BLVM_REQUIRE(dout && dx && L_out > 0 && L_in > 0 && N > 0, "resample_add_bwd: bad arguments");
  hipLaunchKernelGGL(resample_add_bwd_kernel, ew_grid((size_t)L_out * N), dim3(256), 0, static_cast<hipStream_t>(stream_), dout,
                     L_out, L_in, N, dx);
  BLVM_CHECK_LAUNCH("resample_add_bwd");
  return BLVM_OK;
}
