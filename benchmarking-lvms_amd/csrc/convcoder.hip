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

// ---- column statistics in float64 ------------------------------------------------------------------------------------
// mode 0: s1 = sum_t x, s2 = sum_t x^2           (forward statistics)
// mode 1: s1 = sum_t dy, s2 = sum_t dy * xhat    (backward), xhat = (x - mean) * rstd with mr = [mean | rstd]
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ mr, int L, int N, int rows_per_block,
                                                        int mode, double* __restrict__ out) {
  __shared__ double p1[4][64], p2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  double s1 = 0.0, s2 = 0.0;
  if (c < N) {
    if (mode == 0) {
      for (int r = r0 + rl; r < r1; r += 4) {
        const float v = x[(size_t)r * N + c];
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    } else {
      const float mean = mr[c], rstd = mr[N + c];
      for (int r = r0 + rl; r < r1; r += 4) {
        const float g = dy[(size_t)r * N + c];
        const float xh = (x[(size_t)r * N + c] - mean) * rstd;
        s1 += (double)g;
        s2 += (double)g * (double)xh;
      }
    }
  }
  p1[rl][threadIdx.x & 63] = s1;
  p2[rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl == 0 && c < N) {
    const int i = threadIdx.x;
    atomicAdd(out + c, p1[0][i] + p1[1][i] + p1[2][i] + p1[3][i]);
    atomicAdd(out + N + c, p2[0][i] + p2[1][i] + p2[2][i] + p2[3][i]);
  }
}

__global__ __launch_bounds__(256) void finalize_stats_kernel(const double* __restrict__ sums, int L, int N, float eps,
                                                             float* __restrict__ mr) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const double mean = sums[c] / L;
  double var = sums[N + c] / L - mean * mean;  // biased variance, as nn.GroupNorm
  if (var < 0.0) var = 0.0;
  mr[c] = (float)mean;
  mr[N + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = (x - mean) * rstd * gamma_c + beta_c
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         size_t rows, int N, int C, float* __restrict__ y) {
  const int n4 = N / 4;
  const size_t total = rows * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / n4;
    const int n = (int)(i - r * n4) * 4, c = n % C;
    const float4 v = *reinterpret_cast<const float4*>(x + r * N + n);
    const float4 m = *reinterpret_cast<const float4*>(mr + n);
    const float4 s = *reinterpret_cast<const float4*>(mr + N + n);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v.x - m.x) * s.x * g.x + b.x; o.y = (v.y - m.y) * s.y * g.y + b.y;
    o.z = (v.z - m.z) * s.z * g.z + b.z; o.w = (v.w - m.w) * s.w * g.w + b.w;
    *reinterpret_cast<float4*>(y + r * N + n) = o;
  }
}

// dx = gamma_c * rstd * (dy - s1/L - xhat * s2/L)   [optionally * (x > 0): ReLU in front of the norm]
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ mr, const double* __restrict__ sums,
                                                             const float* __restrict__ gamma, size_t rows, int N, int C,
                                                             int relu_mask, float* __restrict__ dx) {
  const size_t total = rows * N;
  const double invL = 1.0 / (double)rows;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / N;
    const int n = (int)(i - r * N), c = n % C;
    const float xv = x[i];
    const float mean = mr[n], rstd = mr[N + n];
    const float xh = (xv - mean) * rstd;
    const float m1 = (float)(sums[n] * invL), m2 = (float)(sums[N + n] * invL);
    float v = gamma[c] * rstd * (dy[i] - m1 - xh * m2);
    if (relu_mask && !(xv > 0.f)) v = 0.f;
    dx[i] = v;
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
// `mask` (optional, shape of x): x is multiplied by (mask > 0) on the fly — the ReLU derivative of the stencil's OUTPUT
// when x is an incoming gradient.  `relu`: apply ReLU to y.
struct DwArgs {
  const float* x;
  const float* mask;
  const float* w;     // [C,k]
  const float* bias;  // [C] or null
  float* y;
  int Lx, Ly, N, C, k, s, d, relu;
};

template <bool SCATTER>
__global__ __launch_bounds__(256) void dw_stencil_kernel(DwArgs a) {
  const int n4 = a.N / 4;
  const size_t total = (size_t)a.Ly * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int u = (int)(i / n4);
    const int n = (int)(i - (size_t)u * n4) * 4, c = n % a.C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) { acc[0] = a.bias[c]; acc[1] = a.bias[c + 1]; acc[2] = a.bias[c + 2]; acc[3] = a.bias[c + 3]; }
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
      float4 v = *reinterpret_cast<const float4*>(a.x + (size_t)r * a.N + n);
      if (a.mask) {
        const float4 m = *reinterpret_cast<const float4*>(a.mask + (size_t)r * a.N + n);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
      }
      acc[0] = fmaf(a.w[(c + 0) * a.k + j], v.x, acc[0]);
      acc[1] = fmaf(a.w[(c + 1) * a.k + j], v.y, acc[1]);
      acc[2] = fmaf(a.w[(c + 2) * a.k + j], v.z, acc[2]);
      acc[3] = fmaf(a.w[(c + 3) * a.k + j], v.w, acc[3]);
    }
    if (a.relu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = acc[q] > 0.f ? acc[q] : 0.f;
    }
    *reinterpret_cast<float4*>(a.y + (size_t)u * a.N + n) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// weight / bias gradient: dw[c,j] += sum_{u,b} A[u,b,c] * Bm[u*s + j*d, b, c], dbias[c] += sum A      (A masked by mask > 0)
// Conv1d: A = dout (mask = out), Bm = in.   ConvTranspose1d: A = in, Bm = dout (mask applies to Bm: see flag).
struct DwWArgs {
  const float *A, *Bm, *mask;
  float *dw, *dbias;
  int LA, LB, N, C, k, s, d, rows_per_block, mask_on_b, bias_from_b;
};

__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWArgs a) {
  __shared__ float part[4][64][9];  // up to 8 taps + bias
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int u0 = blockIdx.y * a.rows_per_block, u1 = min(a.LA, u0 + a.rows_per_block);
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.f;
  if (col < a.N) {
    for (int u = u0 + rl; u < u1; u += 4) {
      float av = a.A[(size_t)u * a.N + col];
      if (a.mask && !a.mask_on_b) av = a.mask[(size_t)u * a.N + col] > 0.f ? av : 0.f;
      if (!a.bias_from_b) acc[8] += av;
      for (int j = 0; j < a.k; ++j) {
        const int r = u * a.s + j * a.d;
        if (r >= a.LB) break;
        float bv = a.Bm[(size_t)r * a.N + col];
        if (a.mask && a.mask_on_b) bv = a.mask[(size_t)r * a.N + col] > 0.f ? bv : 0.f;
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

// masked column sum over (rows, batch) per channel: dbias[c] += sum_{r,b} (mask > 0 ? x : 0)
__global__ __launch_bounds__(256) void masked_chan_sum_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                              int L, int N, int C, int rows_per_block, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(L, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float v = x[(size_t)r * N + col];
      s += (mask == nullptr || mask[(size_t)r * N + col] > 0.f) ? v : 0.f;
    }
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < N) {
    const int i = threadIdx.x;
    atomicAdd(out + col % C, part[0][i] + part[1][i] + part[2][i] + part[3][i]);
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

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_chan_norm_workspace_doubles(int N) { return (size_t)2 * N; }

extern "C" int blvm_chan_norm_fwd(const float* x, int L, int N, int C, const float* gamma, const float* beta, float eps,
                                  float* y, float* mr, double* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && gamma && beta && y && mr && workspace, "chan_norm_fwd: null pointer");
  BLVM_REQUIRE(aligned16(x) && aligned16(y) && aligned16(mr) && aligned16(gamma) && aligned16(beta), "chan_norm_fwd: alignment");
  BLVM_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s));
  int rpb = 64;
  while ((L + rpb - 1) / rpb > 4096) rpb *= 2;
  hipLaunchKernelGGL(col_stats_kernel, dim3((N + 63) / 64, (L + rpb - 1) / rpb), dim3(256), 0, s, x, nullptr, nullptr, L, N, rpb, 0, workspace);
  hipLaunchKernelGGL(finalize_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, s, workspace, L, N, eps, mr);
  hipLaunchKernelGGL(norm_apply_kernel, ew_grid((size_t)L * (N / 4)), dim3(256), 0, s, x, mr, gamma, beta, (size_t)L, N, C, y);
  BLVM_CHECK_LAUNCH("chan_norm_fwd");
  return BLVM_OK;
}

extern "C" int blvm_chan_norm_bwd(const float* x, const float* dy, const float* mr, const float* gamma, int L, int N, int C,
                                  int relu_mask, float* dx, float* dgamma, float* dbeta, double* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && dy && mr && gamma && dx && workspace, "chan_norm_bwd: null pointer");
  BLVM_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s));
  int rpb = 64;
  while ((L + rpb - 1) / rpb > 4096) rpb *= 2;
  hipLaunchKernelGGL(col_stats_kernel, dim3((N + 63) / 64, (L + rpb - 1) / rpb), dim3(256), 0, s, x, dy, mr, L, N, rpb, 1, workspace);
  hipLaunchKernelGGL(norm_bwd_apply_kernel, ew_grid((size_t)L * N), dim3(256), 0, s, x, dy, mr, workspace, gamma, (size_t)L, N, C,
                     relu_mask, dx);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(norm_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, N / C, C, N, dgamma, dbeta);
  BLVM_CHECK_LAUNCH("chan_norm_bwd");
  return BLVM_OK;
}

static int dw_len_out(int L_in, int k, int stride, int dilation, int transposed) {
  const int k_eff = dilation * (k - 1) + 1;
  return transposed ? (L_in - 1) * stride + k_eff : (L_in - k_eff) / stride + 1;
}

extern "C" int blvm_dwconv_out_length(int L_in, int k, int stride, int dilation, int transposed) {
  return dw_len_out(L_in, k, stride, dilation, transposed);
}

extern "C" int blvm_dwconv_fwd(const float* x, const float* w, const float* bias, int L_in, int N, int C, int k, int stride,
                               int dilation, int transposed, int relu, float* y, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L_in, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && w && y && k > 0 && k <= 8 && stride > 0 && dilation > 0, "dwconv_fwd: bad arguments");
  const int L_out = dw_len_out(L_in, k, stride, dilation, transposed);
  BLVM_REQUIRE(L_out > 0, "dwconv_fwd: input of length %d is shorter than the kernel", L_in);
  BLVM_REQUIRE(aligned16(x) && aligned16(y), "dwconv_fwd: alignment");
  DwArgs a{x, nullptr, w, bias, y, L_in, L_out, N, C, k, stride, dilation, relu};
  if (transposed) hipLaunchKernelGGL((dw_stencil_kernel<true>), ew_grid((size_t)L_out * (N / 4)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((dw_stencil_kernel<false>), ew_grid((size_t)L_out * (N / 4)), dim3(256), 0, s, a);
  BLVM_CHECK_LAUNCH("dwconv_fwd");
  return BLVM_OK;
}

extern "C" int blvm_dwconv_bwd(const float* x, const float* w, const float* y, const float* dy, int L_in, int N, int C, int k,
                               int stride, int dilation, int transposed, int relu, float* dx, float* dw, float* dbias,
                               void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_ln(L_in, N, C);
  if (rc) return rc;
  BLVM_REQUIRE(x && w && dy && k > 0 && k <= 8 && stride > 0 && dilation > 0, "dwconv_bwd: bad arguments");
  BLVM_REQUIRE(!relu || y, "dwconv_bwd: the ReLU mask needs the forward output");
  const int L_out = dw_len_out(L_in, k, stride, dilation, transposed);
  const float* mask = relu ? y : nullptr;
  if (dx) {  // data gradient: the other stencil form applied to the (masked) output gradient
    DwArgs a{dy, mask, w, nullptr, dx, L_out, L_in, N, C, k, stride, dilation, 0};
    if (transposed) hipLaunchKernelGGL((dw_stencil_kernel<false>), ew_grid((size_t)L_in * (N / 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dw_stencil_kernel<true>), ew_grid((size_t)L_in * (N / 4)), dim3(256), 0, s, a);
  }
  if (dw) {
    DwWArgs a{};
    a.dw = dw; a.dbias = dbias; a.N = N; a.C = C; a.k = k; a.s = stride; a.d = dilation; a.mask = mask;
    if (!transposed) { a.A = dy; a.Bm = x; a.LA = L_out; a.LB = L_in; a.mask_on_b = 0; a.bias_from_b = 0; }
    else { a.A = x; a.Bm = dy; a.LA = L_in; a.LB = L_out; a.mask_on_b = 1; a.bias_from_b = 1; }
    int rpb = 64;
    while ((a.LA + rpb - 1) / rpb > 4096) rpb *= 2;
    a.rows_per_block = rpb;
    hipLaunchKernelGGL(dw_wgrad_kernel, dim3((N + 63) / 64, (a.LA + rpb - 1) / rpb), dim3(256), 0, s, a);
    if (transposed && dbias) {
      int rb = 64;
      while ((L_out + rb - 1) / rb > 4096) rb *= 2;
      hipLaunchKernelGGL(masked_chan_sum_kernel, dim3((N + 63) / 64, (L_out + rb - 1) / rb), dim3(256), 0, s, dy, mask, L_out, N, C, rb, dbias);
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
