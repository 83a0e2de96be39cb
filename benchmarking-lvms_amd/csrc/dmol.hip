// dmol.hip — K7: fused discretized-mixture-of-logistics head (forward + backward).
//
// One pass over the decoder activations replaces, per audio frame:
//   Linear(F->F)                          blvm/modules/distributions.py:381-382
//   split / clamp(log_scale >= -7)        blvm/modules/distributions.py:383-387
//   10-component DMoL log-likelihood      blvm/utils/log_likelihoods.py:170-231
//   sequence mask + per-utterance sum     blvm/models/vrnn.py:266-269  (float64 accumulation)
// The reference materialises ~15 [B,T,1,10] temporaries for this; here a frame's 30 activations are read once
// (128 B/frame forward, 376 B/frame forward+backward — SURVEY.md §8d) and everything else lives in registers.
//
// Mapping (HBM-bound streaming kernel): one lane per audio frame, 256 consecutive frames per workgroup.  The
// 256 x F activations are fetched with coalesced 16-byte loads into LDS (row pad to F+1 floats: conflict-free
// column reads), each lane then pulls its frame into VGPRs.  The FxF weights are wave-uniform, so hipcc keeps
// them on the scalar path (s_load / SGPR operands of v_fmac).  Per-utterance sums: a wave whose 64 frames
// belong to one utterance (always true for stack sizes that are multiples of 64) reduces with DPP shuffles and
// issues ONE fp64 atomic; otherwise lanes fall back to per-lane atomics.
#include "common.h"

namespace blvm {
namespace {

constexpr int F_MAX = 30;   // 3 * num_mix, num_mix = 10
constexpr int NMIX = 10;
constexpr int FPB = 256;    // frames per block

struct DmolArgs {
  const float* dec;
  const float* W;
  const float* bias;
  const float* y;
  const int32_t* x_sl;
  const float* g_b;
  double* log_prob;
  float* ll_twise;
  float* d_dec;
  float* d_par;
  long long n_frames;  // rows * S
  int layout, B, T, Tp, S;
  float half_bin, low_edge, high_edge, log_half_bins, log_eps;
};

// Stage 256 frames x 30 floats (contiguous in HBM) into LDS [256][31].
__device__ __forceinline__ void stage_frames(const float* __restrict__ src, long long f0, long long n_frames,
                                             float* __restrict__ lds) {
  const long long base = f0 * F_MAX;
  const long long total = min((long long)FPB, n_frames - f0) * F_MAX;  // floats available
  const bool vec = ((reinterpret_cast<uintptr_t>(src + base)) & 15u) == 0;
  for (int i = threadIdx.x * 4; i < FPB * F_MAX; i += 256 * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < total && vec) {
      v = *reinterpret_cast<const float4*>(src + base + i);
    } else {
      if (i + 0 < total) v.x = src[base + i + 0];
      if (i + 1 < total) v.y = src[base + i + 1];
      if (i + 2 < total) v.z = src[base + i + 2];
      if (i + 3 < total) v.w = src[base + i + 3];
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = i + k, fr = idx / F_MAX, c = idx - fr * F_MAX;
      lds[fr * (F_MAX + 1) + c] = e[k];
    }
  }
}

// Write 256 frames x 30 floats from LDS [256][31] back to HBM with coalesced 16-byte stores.
__device__ __forceinline__ void unstage_frames(float* __restrict__ dst, long long f0, long long n_frames,
                                               const float* __restrict__ lds) {
  const long long base = f0 * F_MAX;
  const long long total = min((long long)FPB, n_frames - f0) * F_MAX;
  const bool vec = ((reinterpret_cast<uintptr_t>(dst + base)) & 15u) == 0;
  for (int i = threadIdx.x * 4; i < FPB * F_MAX; i += 256 * 4) {
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = i + k, fr = idx / F_MAX, c = idx - fr * F_MAX;
      e[k] = lds[fr * (F_MAX + 1) + c];
    }
    if (i + 3 < total && vec) {
      *reinterpret_cast<float4*>(dst + base + i) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < total) dst[base + i + k] = e[k];
    }
  }
}

struct FrameCoord {
  int b;
  int tau;
  bool valid;
};

__device__ __forceinline__ FrameCoord frame_coord(const DmolArgs& a, long long f) {
  FrameCoord c;
  c.b = 0; c.tau = 0; c.valid = false;
  if (f >= a.n_frames) return c;
  const long long row = f / a.S;
  const int j = (int)(f - row * a.S);
  int t;
  if (a.layout == 0) { c.b = (int)(row / a.Tp); t = (int)(row - (long long)c.b * a.Tp); }
  else { t = (int)(row / a.B); c.b = (int)(row - (long long)t * a.B); }
  c.tau = t * a.S + j;
  c.valid = (c.tau < a.T) && (c.tau < a.x_sl[c.b]);
  return c;
}

// Per-frame DMoL math.  p[0..9] logits, p[10..19] locs, p[20..29] raw log-scales.
// Returns ll; if BWD, overwrites p with dll/dp.
template <bool BWD>
__device__ __forceinline__ float dmol_frame(const DmolArgs& a, float yv, float (&p)[F_MAX]) {
  float mx = p[0];
#pragma unroll
  for (int m = 1; m < NMIX; ++m) mx = fmaxf(mx, p[m]);
  float se = 0.f;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) se += exp_(p[m] - mx);
  const float lse_logits = mx + log_(se);

  float lp[NMIX], dloc[NMIX], dls[NMIX];
  const bool is_low = yv < a.low_edge, is_high = yv > a.high_edge;
  float tmax = -INFINITY;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) {
    const float loc = p[NMIX + m];
    const float raw = p[2 * NMIX + m];
    const float ls = fmaxf(raw, a.log_eps);
    const float c = yv - loc;
    const float inv = exp_(-ls);
    const float plus = inv * (c + a.half_bin);
    const float minus = inv * (c - a.half_bin);
    const float sp = sigmoidf_(plus), sm = sigmoidf_(minus);
    const float delta = sp - sm;
    float v, gl, gs;  // value, d/dloc, d/dls
    if (is_high) {
      v = -softplusf_(minus);
      if (BWD) { gl = inv * sm; gs = minus * sm; }
    } else if (is_low) {
      v = plus - softplusf_(plus);
      if (BWD) { const float q = 1.f - sp; gl = -inv * q; gs = -plus * q; }
    } else if (delta > 1e-5f) {
      v = log_(fmaxf(delta, 1e-10f));
      if (BWD) {
        const float dp = sp * (1.f - sp), dm = sm * (1.f - sm), rd = 1.f / delta;
        gl = -inv * (dp - dm) * rd;
        gs = -(plus * dp - minus * dm) * rd;
      }
    } else {
      const float mid = inv * c;
      v = mid - ls - 2.f * softplusf_(mid) - a.log_half_bins;
      if (BWD) { const float q = 1.f - 2.f * sigmoidf_(mid); gl = -inv * q; gs = -mid * q - 1.f; }
    }
    lp[m] = v + (p[m] - lse_logits);
    tmax = fmaxf(tmax, lp[m]);
    if (BWD) { dloc[m] = gl; dls[m] = (raw >= a.log_eps) ? gs : 0.f; }
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) s += exp_(lp[m] - tmax);
  const float ll = tmax + log_(s);
  if (BWD) {
    const float rs = 1.f / s, rse = 1.f / se;
#pragma unroll
    for (int m = 0; m < NMIX; ++m) {
      const float w = exp_(lp[m] - tmax) * rs;            // responsibility of component m
      const float pm = exp_(p[m] - mx) * rse;             // softmax(logits)_m
      p[m] = w - pm;
      p[NMIX + m] = w * dloc[m];
      p[2 * NMIX + m] = w * dls[m];
    }
  }
  return ll;
}

template <bool BWD>
__global__ __launch_bounds__(256) void dmol_kernel(DmolArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[FPB * (F_MAX + 1)];
  const long long f0 = (long long)blockIdx.x * FPB;
  const long long f = f0 + threadIdx.x;
  stage_frames(a.dec, f0, a.n_frames, lds);
  __syncthreads();

  const FrameCoord fc = frame_coord(a, f);
  float d[F_MAX];
#pragma unroll
  for (int i = 0; i < F_MAX; ++i) d[i] = lds[threadIdx.x * (F_MAX + 1) + i];

  // p = W d + bias   (weights are wave-uniform -> scalar operands); W == NULL: the parameters ARE the input
  float p[F_MAX];
  if (a.W != nullptr) {
#pragma unroll
    for (int o = 0; o < F_MAX; ++o) {
      float s = a.bias[o];
#pragma unroll
      for (int i = 0; i < F_MAX; ++i) s = fmaf(a.W[o * F_MAX + i], d[i], s);
      p[o] = s;
    }
  } else {
#pragma unroll
    for (int o = 0; o < F_MAX; ++o) p[o] = d[o];
  }
  const float yv = fc.valid ? a.y[(size_t)fc.b * a.T + fc.tau] : 0.f;
  const float ll = dmol_frame<BWD>(a, yv, p);

  if (!BWD) {
    const float llm = fc.valid ? ll : 0.f;
    if (a.ll_twise != nullptr && fc.valid) a.ll_twise[(size_t)fc.b * a.T + fc.tau] = llm;
    // per-utterance fp64 sums
    const int b0 = __builtin_amdgcn_readfirstlane(fc.b);
    const bool uniform = __all((fc.b == b0) || !fc.valid);
    if (uniform) {
      double v = (double)llm;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      // lane 0 may itself be invalid (b0 arbitrary then): pick b from any valid lane
      const unsigned long long vm = __ballot(fc.valid);
      if (vm != 0ull) {
        const int src = __ffsll((long long)vm) - 1;
        const int bb = __shfl(fc.b, src, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.log_prob + bb, v);
      }
    } else if (fc.valid) {
      atomicAdd(a.log_prob + fc.b, (double)llm);
    }
  } else {
    const float g = fc.valid ? a.g_b[fc.b] : 0.f;
    // p now holds dll/dp; scale by upstream
#pragma unroll
    for (int o = 0; o < F_MAX; ++o) p[o] *= g;
    // d_dec = W^T dp
    float dd[F_MAX];
    if (a.W != nullptr) {
#pragma unroll
      for (int i = 0; i < F_MAX; ++i) dd[i] = 0.f;
#pragma unroll
      for (int o = 0; o < F_MAX; ++o) {
#pragma unroll
        for (int i = 0; i < F_MAX; ++i) dd[i] = fmaf(a.W[o * F_MAX + i], p[o], dd[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < F_MAX; ++i) dd[i] = p[i];
    }
    __syncthreads();  // everyone has read its frame from lds
#pragma unroll
    for (int i = 0; i < F_MAX; ++i) lds[threadIdx.x * (F_MAX + 1) + i] = dd[i];
    __syncthreads();
    unstage_frames(a.d_dec, f0, a.n_frames, lds);
    if (a.d_par != nullptr) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < F_MAX; ++i) lds[threadIdx.x * (F_MAX + 1) + i] = p[i];
      __syncthreads();
      unstage_frames(a.d_par, f0, a.n_frames, lds);
    }
  }
}

int check_common(const float* dec, const float* W, const float* bias, const float* y, const int32_t* x_sl, int B,
                 int T, int Tp, int S, int num_mix, int num_bins, int layout) {
  BLVM_REQUIRE(dec && y && x_sl, "dmol: null pointer");
  BLVM_REQUIRE((W == nullptr) == (bias == nullptr), "dmol: W and bias must both be given or both be NULL");
  BLVM_REQUIRE(num_mix == NMIX, "dmol: this build supports num_mix == 10 (got %d)", num_mix);
  BLVM_REQUIRE(B > 0 && T > 0 && Tp > 0 && S > 0 && num_bins > 1, "dmol: bad shape B=%d T=%d Tp=%d S=%d", B, T, Tp, S);
  BLVM_REQUIRE((long long)Tp * S >= T, "dmol: Tp*S (%d*%d) < T (%d)", Tp, S, T);
  BLVM_REQUIRE(layout == 0 || layout == 1, "dmol: layout must be 0 or 1");
  return BLVM_OK;
}

DmolArgs make_args(const float* dec, int layout, const float* W, const float* bias, const float* y,
                   const int32_t* x_sl, int B, int T, int Tp, int S, int num_bins, float log_eps) {
  DmolArgs a{};
  a.dec = dec; a.W = W; a.bias = bias; a.y = y; a.x_sl = x_sl;
  a.n_frames = (long long)B * Tp * S;
  a.layout = layout; a.B = B; a.T = T; a.Tp = Tp; a.S = S;
  a.half_bin = 1.0f / (float)(num_bins - 1);
  a.low_edge = (float)(2.0 / num_bins - 1.0);
  a.high_edge = (float)(1.0 - 2.0 / num_bins);
  a.log_half_bins = (float)log((double)num_bins / 2.0);
  a.log_eps = log_eps;
  return a;
}

}  // namespace
}  // namespace blvm

extern "C" int blvm_dmol_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                             const int32_t* x_sl, int B, int T, int Tp, int S, int num_mix, int num_bins,
                             float log_eps, double* log_prob, float* ll_twise, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, num_mix, num_bins, layout);
  if (rc) return rc;
  BLVM_REQUIRE(log_prob != nullptr, "dmol_fwd: log_prob is null");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, num_bins, log_eps);
  a.log_prob = log_prob;
  a.ll_twise = ll_twise;
  const long long blocks = (a.n_frames + FPB - 1) / FPB;
  BLVM_REQUIRE(blocks < (1ll << 31), "dmol_fwd: too many frames");
  hipLaunchKernelGGL((dmol_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("dmol_fwd");
  return BLVM_OK;
}

extern "C" int blvm_dmol_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                             const int32_t* x_sl, const float* g_b, int B, int T, int Tp, int S, int num_mix,
                             int num_bins, float log_eps, float* d_dec, float* d_par, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, num_mix, num_bins, layout);
  if (rc) return rc;
  BLVM_REQUIRE(g_b && d_dec && (d_par || !W), "dmol_bwd: null pointer");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, num_bins, log_eps);
  a.g_b = g_b;
  a.d_dec = d_dec;
  a.d_par = d_par;
  const long long blocks = (a.n_frames + FPB - 1) / FPB;
  BLVM_REQUIRE(blocks < (1ll << 31), "dmol_bwd: too many frames");
  hipLaunchKernelGGL((dmol_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("dmol_bwd");
  return BLVM_OK;
}
