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

// The 30x30 head weights are read-only for the whole launch and wave-uniform: reading them through the CONSTANT
// address space makes hipcc keep them on the scalar path (s_load -> SGPR operands of v_pk_fma) even inside loops that
// also store to global memory (otherwise it falls back to 900 vector loads of a uniform address per iteration).
typedef const __attribute__((address_space(4))) float cfloat;
__device__ __forceinline__ cfloat* as_const(const float* p) { return (cfloat*)(uintptr_t)p; }

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
  int kind;                 // 0 discretized logistic mixture, 1 Gaussian mixture
  float sd_beta, sd_eps;    // kind 1: sd = softplus_beta(raw) + sd_eps
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

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): the kernel has a budget of ~1700 vector
// instructions per frame before it stops being HBM-bound; libm-accurate expf/logf/log1pf/division cost 4x that.
// (the bare instructions: `__expf` / `__logf` expand to a range-scaled sequence — compare, select, v_ldexp — around v_exp_f32 /
// v_log_f32 for denormal results and arguments, ~7 instructions per call and ~60 calls per frame; here results below 2^-126 may
// flush to zero, where they are added to sums >= 1, and every log argument is >= 1e-10)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Per-frame DMoL math.  p[0..9] logits, p[10..19] locs, p[20..29] raw log-scales.
// Returns ll; if BWD, overwrites p with dll/dp.
//
// Same quantities as blvm/utils/log_likelihoods.py:170-231, evaluated without the fp32 cancellation of
// sigmoid(plus) - sigmoid(minus):  with m = (y - loc)/s, w = half_bin/s, t = exp(-|m|)
//     delta = sigma(m+w) - sigma(m-w) = sinh(w) / (cosh(m) + cosh(w)) = 2 t sinh(w) / (1 + t^2 + 2 t cosh(w))
//     log_pdf_mid = m - log s - 2 softplus(m),  softplus(m) = max(m,0) + log(1 + t)
//     d log(delta)/dm = -sinh(m)/(cosh(m)+cosh(w)) = -sign(m)(1 - t^2)/(1 + t^2 + 2 t cosh(w)),   1 - 2 sigma(m) = -sign(m)(1-t)/(1+t)
// (2 exp + 1 log + 2 rcp per component).  The edge bins (|y| > 1 - 2/bins) are rare and take the libm path.
template <bool BWD>
__device__ __forceinline__ float dmol_frame(const DmolArgs& a, float yv, float (&p)[F_MAX]) {
  float mx = p[0];
#pragma unroll
  for (int m = 1; m < NMIX; ++m) mx = fmaxf(mx, p[m]);
  float se = 0.f;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) se += fexp(p[m] - mx);
  const float lse_logits = mx + flog(se);

  float lp[NMIX], dloc[NMIX], dls[NMIX];
  const bool is_low = yv < a.low_edge, is_high = yv > a.high_edge;
  const bool any_edge = __any(is_low || is_high);  // wave-uniform
  float tmax = -INFINITY;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) {
    const float loc = p[NMIX + m];
    const float raw = p[2 * NMIX + m];
    const float ls = fmaxf(raw, a.log_eps);
    const float inv = fexp(-ls);
    const float mm = inv * (yv - loc);
    const float w = inv * a.half_bin;
    const float t = fexp(-fabsf(mm));
    float sh, ch;  // sinh(w), cosh(w); w > 0
    if (w < 0.25f) {
      const float w2 = w * w;
      sh = w * (1.f + w2 * (1.f / 6.f + w2 * (1.f / 120.f)));
      ch = 1.f + w2 * (0.5f + w2 * (1.f / 24.f + w2 * (1.f / 720.f)));
    } else {
      const float ew = fexp(w), iw = frcp(ew);
      sh = 0.5f * (ew - iw);
      ch = 0.5f * (ew + iw);
    }
    const float rden = frcp(1.f + t * t + 2.f * t * ch);
    const float delta = 2.f * t * sh * rden;
    const bool big = delta > 1e-5f;
    const float L = flog(big ? fmaxf(delta, 1e-10f) : 1.f + t);
    float v = big ? L : mm - ls - 2.f * fmaxf(mm, 0.f) - 2.f * L - a.log_half_bins;
    float gl = 0.f, gs = 0.f;
    if (BWD) {
      const float sgn = mm < 0.f ? -1.f : 1.f;
      if (big) {
        const float s_over = sgn * (1.f - t * t) * rden;  // sinh(m) / (cosh(m) + cosh(w))
        gl = inv * s_over;
        gs = mm * s_over - w * ch * frcp(sh) + w * delta;
      } else {
        const float q = -sgn * (1.f - t) * frcp(1.f + t);  // 1 - 2 sigmoid(m)
        gl = -inv * q;
        gs = -mm * q - 1.f;
      }
    }
    if (any_edge) {
      const float plus = mm + w, minus = mm - w;
      if (is_high) {
        v = -softplusf_(minus);
        if (BWD) { const float sm = sigmoidf_(minus); gl = inv * sm; gs = minus * sm; }
      } else if (is_low) {
        v = plus - softplusf_(plus);
        if (BWD) { const float q = 1.f - sigmoidf_(plus); gl = -inv * q; gs = -plus * q; }
      }
    }
    lp[m] = v + (p[m] - lse_logits);
    tmax = fmaxf(tmax, lp[m]);
    if (BWD) { dloc[m] = gl; dls[m] = (raw >= a.log_eps) ? gs : 0.f; }
  }
  float s = 0.f;
  float e[NMIX];
#pragma unroll
  for (int m = 0; m < NMIX; ++m) { e[m] = fexp(lp[m] - tmax); s += e[m]; }
  const float ll = tmax + flog(s);
  if (BWD) {
    const float rs = frcp(s), rse = frcp(se);
#pragma unroll
    for (int m = 0; m < NMIX; ++m) {
      const float wgt = e[m] * rs;                     // responsibility of component m
      const float pm = fexp(p[m] - mx) * rse;          // softmax(logits)_m
      p[m] = wgt - pm;
      p[NMIX + m] = wgt * dloc[m];
      p[2 * NMIX + m] = wgt * dls[m];
    }
  }
  return ll;
}

// Per-frame Gaussian-mixture math (`DiagonalGaussianMixtureDense.forward` blvm/modules/distributions.py:190-206 after its
// Linear; `gaussian_mixture_ll` blvm/utils/log_likelihoods.py:42-60 with epsilon = 0): p[0..9] logits, p[10..19] means,
// p[20..29] pre-softplus standard deviations.  Returns ll; if BWD, overwrites p with dll/dp.
template <bool BWD>
__device__ __forceinline__ float gmm_frame(const DmolArgs& a, float yv, float (&p)[F_MAX]) {
  float mx = p[0];
#pragma unroll
  for (int m = 1; m < NMIX; ++m) mx = fmaxf(mx, p[m]);
  float se = 0.f;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) se += fexp(p[m] - mx);
  const float lse_logits = mx + flog(se);
  float lp[NMIX], dmu[NMIX], dsd[NMIX];
  float tmax = -INFINITY;
  const float inv_beta = 1.f / a.sd_beta;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) {
    const float mu = p[NMIX + m], raw = p[2 * NMIX + m];
    const float sd = softplus_beta(raw, a.sd_beta, inv_beta) + a.sd_eps;
    const float isd = frcp(sd), d = (yv - mu) * isd;
    lp[m] = -0.5f * d * d - flog(sd) - 0.91893853320467274f + (p[m] - lse_logits);
    tmax = fmaxf(tmax, lp[m]);
    if (BWD) {
      dmu[m] = d * isd;                                             // d/dmu
      dsd[m] = (d * d - 1.f) * isd * sigmoidf_(a.sd_beta * raw);    // d/dsd * d sd/d raw
    }
  }
  float s = 0.f, e[NMIX];
#pragma unroll
  for (int m = 0; m < NMIX; ++m) { e[m] = fexp(lp[m] - tmax); s += e[m]; }
  const float ll = tmax + flog(s);
  if (BWD) {
    const float rs = frcp(s), rse = frcp(se);
#pragma unroll
    for (int m = 0; m < NMIX; ++m) {
      const float wgt = e[m] * rs;
      const float pm = fexp(p[m] - mx) * rse;
      p[m] = wgt - pm;
      p[NMIX + m] = wgt * dmu[m];
      p[2 * NMIX + m] = wgt * dsd[m];
    }
  }
  return ll;
}

template <bool BWD>
__device__ __forceinline__ float head_frame(const DmolArgs& a, float yv, float (&p)[F_MAX]) {
  return a.kind == 0 ? dmol_frame<BWD>(a, yv, p) : gmm_frame<BWD>(a, yv, p);  // wave-uniform branch
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
  cfloat* Wc = as_const(a.W);
  cfloat* bc = as_const(a.bias);
  float p[F_MAX];
  if (a.W != nullptr) {
#pragma unroll
    for (int o = 0; o < F_MAX; ++o) {
      float s = bc[o];
#pragma unroll
      for (int i = 0; i < F_MAX; ++i) s = fmaf(Wc[o * F_MAX + i], d[i], s);
      p[o] = s;
    }
  } else {
#pragma unroll
    for (int o = 0; o < F_MAX; ++o) p[o] = d[o];
  }
  const float yv = fc.valid ? a.y[(size_t)fc.b * a.T + fc.tau] : 0.f;
  const float ll = head_frame<BWD>(a, yv, p);

  if (!BWD) {
    const float llm = fc.valid ? ll : 0.f;
    if (a.ll_twise != nullptr && fc.valid) a.ll_twise[(size_t)fc.b * a.T + fc.tau] = llm;
    // per-utterance fp64 sums: wave-uniform utterance -> shuffle reduction + one atomic per wave; otherwise (time-major
    // single-frame rows interleave the whole batch inside a wave) reduce per utterance in LDS first, so that the global
    // fp64 atomics are one per (workgroup, utterance) instead of one per frame on B hot addresses.
    __shared__ double part[256];
    const bool lds_path = a.B <= 256;
    if (lds_path) {
      part[threadIdx.x] = 0.0;
      __syncthreads();
    }
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
        if ((threadIdx.x & 63) == 0) {
          if (lds_path) atomicAdd(&part[bb], v);
          else atomicAdd(a.log_prob + bb, v);
        }
      }
    } else if (fc.valid) {
      if (lds_path) atomicAdd(&part[fc.b], (double)llm);
      else atomicAdd(a.log_prob + fc.b, (double)llm);
    }
    if (lds_path) {
      __syncthreads();
      if ((int)threadIdx.x < a.B && part[threadIdx.x] != 0.0) atomicAdd(a.log_prob + threadIdx.x, part[threadIdx.x]);
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
        for (int i = 0; i < F_MAX; ++i) dd[i] = fmaf(Wc[o * F_MAX + i], p[o], dd[i]);
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

// ---------------------------------------------------------------------------------------------------------------
// Row-wise variant for stack sizes that are multiples of 64 (the VRNN / SRNN / LSTM / STCN decoders, S = 64 or 256).
// A workgroup owns ONE utterance and a chunk of its 64-frame units; each of its four waves walks every fourth unit on its own —
// private LDS image, no workgroup barrier inside the loop — carrying the per-utterance sum in registers (ONE fp64 atomic per
// workgroup; the flat kernel above issues one per wave).  The grid fills the chip once (resident workgroups per CU by registers
// x CUs): a wave's prologue (lengths, the weight table, its first unit) is two dependent HBM round trips, so few long-lived waves
// beat many short ones (2 048 workgroups of 2 units per wave: 40 % of all wave cycles parked).  Both products with the head's
// 30 x 30 Linear (forward p = W d + b; backward d_dec = d_par W) run on the matrix pipe, in place in the image.  Round 3, same
// shapes [64,16000]: forward 55 -> 42 us (3.0 TB/s of the 124 B/frame), backward 125 -> 93 us (3.9 TB/s of 364 B/frame).  What is
// left is arithmetic, not HBM: per unit ~640 vector instructions of mixture math (19 us of a SIMD's time per launch) plus 30 (60)
// fp32 MFMAs (12.5 us per product pass) that do NOT overlap — SQ_VALU_MFMA_COEXEC_CYCLES is 0: the fp32 matrix instructions share
// the vector issue — against 20 us (58 us) of HBM time.  Integer divisions are 32-bit and wave-uniform.
// ---------------------------------------------------------------------------------------------------------------
// A unit = the 64 frames x 30 floats (7 680 bytes, contiguous and 16-byte aligned in HBM) one wave works on.  Its LDS image is a
// VERBATIM copy (no padding, no index arithmetic: 480 float4, lane q -> float4 q), private to the wave:
//   * the lane's own frame is 15 ds_read_b64 at byte 120 * lane (dword stride 30: within each half-wave of a b64 access the 32
//     lanes fall on 32 distinct even banks — 15 is invertible mod 32 — so the reads are conflict-free),
//   * the matrix pipe's A operand IN[32 t + li][2 j + lh] is dword 30 (32 t + li) + 2 j + lh: 64 distinct banks again,
//   * nothing but this wave touches the image, so the only ordering needed is the wave's own program order (LDS executes a
//     wave's instructions in order): `wave_sync` is a compiler fence, not a barrier — the four waves of a workgroup drift apart,
//     one wave's MFMAs run under another's exp / log stream.
constexpr int UNIT_FLOATS = 64 * F_MAX;  // 1920
// (a compiler-only fence: a release / acquire fence pair, even at wavefront scope, made hipcc wait for ALL outstanding vector
// memory — the next unit's prefetch included — in front of every image phase)
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }
// a unit in flight from HBM: eight plain vector registers (f32x4 k holds float4 lane + 64 k; k = 7: lanes 0..31).  Separate
// scalars of an ext-vector type on purpose: an array of HIP's float4 (a struct of unions) passed by reference stayed in scratch.
#define UNIT_REGS(n) f32x4 n##0, n##1, n##2, n##3, n##4, n##5, n##6, n##7
#define LOAD_UNIT(n, src, lane)                                                                                     \
  do {                                                                                                              \
    const f32x4* s4_ = reinterpret_cast<const f32x4*>(src);                                                         \
    n##0 = s4_[(lane)]; n##1 = s4_[(lane) + 64]; n##2 = s4_[(lane) + 128]; n##3 = s4_[(lane) + 192];                \
    n##4 = s4_[(lane) + 256]; n##5 = s4_[(lane) + 320]; n##6 = s4_[(lane) + 384];                                   \
    n##7 = s4_[448 + ((lane) & 31)]; /* unconditional: lanes 32..63 re-read what lanes 0..31 read and never use it */ \
  } while (0)
#define PUT_UNIT(n, lds, lane)                                                                                      \
  do {                                                                                                              \
    f32x4* d4_ = reinterpret_cast<f32x4*>(lds);                                                                     \
    d4_[(lane)] = n##0; d4_[(lane) + 64] = n##1; d4_[(lane) + 128] = n##2; d4_[(lane) + 192] = n##3;                \
    d4_[(lane) + 256] = n##4; d4_[(lane) + 320] = n##5; d4_[(lane) + 384] = n##6;                                   \
    if ((lane) < 32) d4_[448 + (lane)] = n##7;                                                                      \
  } while (0)
__device__ __forceinline__ void store_unit(float* __restrict__ dst, const float* __restrict__ lds, int lane) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(lds);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int k = 0; k < 7; ++k) d4[lane + 64 * k] = s4[lane + 64 * k];
  if (lane < 32) d4[448 + lane] = s4[448 + lane];
}

// A wave's 64 x 30 image times a 30 x 30 matrix on the matrix pipe, IN PLACE: OUT[f][n] = sum_k IN[f][k] M[k][n] (+ bias[n]), two
// 32-row tiles, K = 30 as 15 k-pairs of v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain, k ascending) instead of 900 VALU FMAs per
// frame.  lane (li = lane & 31, lh = lane >> 5): A = IN[32 t + li][2 j + lh], B = M[2 j + lh][li] (0 for li >= 30) from a per-lane
// table in LDS.  A tile's 15 A reads are issued before its 16 result writes and only touch its own rows.
__device__ __forceinline__ void mfma_image_30x30(float* __restrict__ lds, const float* __restrict__ mlds, float bias_n, int lane) {
  const int li = lane & 31, lh = lane >> 5;
  float mreg[15];  // this lane's B operands, from the workgroup's LDS copy [15][64] (30 resident registers per wave cost a wave of occupancy)
#pragma unroll
  for (int j = 0; j < 15; ++j) mreg[j] = mlds[j * 64 + lane];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* row = lds + (32 * t + li) * F_MAX + lh;
    float av[15];  // all fifteen A values requested before the first MFMA (read -> wait -> MFMA pairs expose the LDS latency 15 times)
#pragma unroll
    for (int j = 0; j < 15; ++j) av[j] = row[2 * j];
    wave_sync();  // (all lanes' A reads are issued before any result lands in the image)
#pragma unroll
    for (int j = 0; j < 15; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], mreg[j], acc, 0, 0, 0);
    if (li < F_MAX) {
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[(32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh) * F_MAX + li] = acc[r] + bias_n;
    }
  }
}

template <bool BWD, bool PF>  // PF: the next unit is requested into registers (32 of them) before this one is worked on
__global__ __launch_bounds__(256) void dmol_rows_kernel(DmolArgs a, int units, int nchunks) {
  __shared__ __attribute__((aligned(16))) float lds_all[4 * UNIT_FLOATS];
  __shared__ double wsum[4];
  __shared__ float wtab[2 * 15 * 64];  // the head's Linear as MFMA B operands per lane: [0] forward (W^T), [1] backward (W)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* lds = lds_all + wave * UNIT_FLOATS;
  const int b = blockIdx.x / nchunks, c = blockIdx.x - b * nchunks;
  const int u_begin = (int)((long long)c * units / nchunks), u_end = (int)((long long)(c + 1) * units / nchunks);
  const int upr = a.S >> 6;  // 64-frame units per row
  const int len = min(a.x_sl[b], a.T);
  const float g = BWD ? a.g_b[b] : 0.f;
  const bool lin = a.W != nullptr;  // uniform
  auto unit_base = [&](int u) {  // float offset of unit u of utterance b (wave-uniform)
    const int t = u / upr, j0 = (u - t * upr) * 64;
    const long long row = a.layout == 0 ? (long long)b * a.Tp + t : (long long)t * a.B + b;
    return ((size_t)row * a.S + j0) * F_MAX;
  };
  UNIT_REGS(nxt);
  int u = u_begin + wave;
  if (PF && u < u_end) LOAD_UNIT(nxt, a.dec + unit_base(u), lane);  // the first unit travels with the weights
  // the head's Linear for the matrix pipe: forward B operand W^T[k][n] = W[n][k]; backward (d_dec = d_par W) B operand W[k][n]
  float bias_n = 0.f;
  {
    const int li = lane & 31, lh = lane >> 5;
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        wtab[j * 64 + lane] = (lin && li < F_MAX) ? a.W[li * F_MAX + 2 * j + lh] : 0.f;
        if (BWD) wtab[(15 + j) * 64 + lane] = (lin && li < F_MAX) ? a.W[(2 * j + lh) * F_MAX + li] : 0.f;
      }
    }
    if (lin && li < F_MAX) bias_n = a.bias[li];
  }
  __syncthreads();
  double acc = 0.0;
  for (; u < u_end; u += 4) {
    const int t = u / upr, j0 = (u - t * upr) * 64;
    const size_t base = unit_base(u);
    const int tau = t * a.S + j0 + lane;
    const bool valid = tau < len;
    const float yv = valid ? a.y[(size_t)b * a.T + tau] : 0.f;
    wave_sync();  // the previous unit's last image reads (its stores to HBM) are issued
    if (!PF) LOAD_UNIT(nxt, a.dec + base, lane);
    PUT_UNIT(nxt, lds, lane);
    if (PF && u + 4 < u_end) LOAD_UNIT(nxt, a.dec + unit_base(u + 4), lane);  // the next unit travels while this one is worked on
    wave_sync();
    if (lin) {
      mfma_image_30x30(lds, wtab, bias_n, lane);  // p = W d + bias, in place
      wave_sync();
    }
    float p[F_MAX];
    {
      const float2* fp = reinterpret_cast<const float2*>(lds + lane * F_MAX);
#pragma unroll
      for (int i = 0; i < F_MAX / 2; ++i) { const float2 q = fp[i]; p[2 * i] = q.x; p[2 * i + 1] = q.y; }
    }
    const float ll = head_frame<BWD>(a, yv, p);
    if (!BWD) {
      if (valid) {
        acc += (double)ll;
        if (a.ll_twise != nullptr) a.ll_twise[(size_t)b * a.T + tau] = ll;
      }
    } else {
      const float gv = valid ? g : 0.f;
      wave_sync();  // every lane has its frame: the image may be overwritten
      {
        float2* fp = reinterpret_cast<float2*>(lds + lane * F_MAX);
#pragma unroll
        for (int i = 0; i < F_MAX / 2; ++i) fp[i] = make_float2(p[2 * i] * gv, p[2 * i + 1] * gv);
      }
      wave_sync();
      if (a.d_par != nullptr) {
        store_unit(a.d_par + base, lds, lane);      // d(loss)/d(head output): what the weight / bias gradient GEMM reads
        wave_sync();
      }
      if (lin) {
        mfma_image_30x30(lds, wtab + 15 * 64, 0.f, lane);  // d_dec = d_par W, in place
        wave_sync();
      }
      store_unit(a.d_dec + base, lds, lane);
    }
  }
  if (!BWD) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.log_prob + b, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
}

template <bool BWD>
int launch_dmol(DmolArgs& a, hipStream_t s) {
  if (a.S % 64 == 0 && aligned16(a.dec) && (!BWD || (aligned16(a.d_dec) && (a.d_par == nullptr || aligned16(a.d_par))))) {
    const long long units = (long long)a.Tp * (a.S / 64);
    BLVM_REQUIRE(units < (1ll << 31), "dmol: too many frames");
    // Few, long-lived workgroups: a wave's prologue (x_sl, 30 weight registers, its first unit: two dependent HBM round trips) is
    // ~4 us, a unit ~1.5 us of its time — at 2 units per wave (2048 workgroups) the waves spent 40 % of their lives parked.  About
    // three workgroups per CU (what the registers allow), every wave walking its chunk with the next unit in flight.
    int nchunks = (int)((units + 3) / 4);
    // PF (the next unit requested into 32 registers ahead of time) costs a wave of occupancy in both kernels and measured no faster
    // than one more resident wave per SIMD covering the load (same box: forward 44.8 vs 42.3 us, backward 95.6 vs 92.7): off
    static const int pf = [] { const char* e = getenv("BLVM_DMOL_PF"); return e ? atoi(e) : 0; }();
    static const int wg_env = [] { const char* e = getenv("BLVM_DMOL_WGS"); return e ? atoi(e) : 0; }();
    static int wg_fill[2] = {0, 0};  // workgroups that fill the chip once: resident workgroups per CU (by registers) x CUs
    if (wg_fill[pf ? 1 : 0] == 0) {
      int per_cu = 0, dev = 0, cus = 0;
      const void* k = pf ? reinterpret_cast<const void*>(&dmol_rows_kernel<BWD, true>) : reinterpret_cast<const void*>(&dmol_rows_kernel<BWD, false>);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
      (void)hipGetLastError();
      wg_fill[pf ? 1 : 0] = per_cu * cus;
    }
    const int wg_target = wg_env > 0 ? wg_env : wg_fill[pf ? 1 : 0];
    const int cap = (wg_target + a.B - 1) / a.B;
    if (nchunks > cap) nchunks = cap;
    if (nchunks < 1) nchunks = 1;
    if (pf) hipLaunchKernelGGL((dmol_rows_kernel<BWD, true>), dim3((unsigned)(a.B * nchunks)), dim3(256), 0, s, a, (int)units, nchunks);
    else hipLaunchKernelGGL((dmol_rows_kernel<BWD, false>), dim3((unsigned)(a.B * nchunks)), dim3(256), 0, s, a, (int)units, nchunks);
  } else {
    const long long blocks = (a.n_frames + FPB - 1) / FPB;
    BLVM_REQUIRE(blocks < (1ll << 31), "dmol: too many frames");
    hipLaunchKernelGGL((dmol_kernel<BWD>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  }
  return BLVM_OK;
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
  rc = launch_dmol<false>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
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
  rc = launch_dmol<true>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("dmol_bwd");
  return BLVM_OK;
}

namespace blvm {
namespace {

// ---- single diagonal Gaussian head (`DiagonalGaussianDense` as a likelihood, blvm/modules/distributions.py:105-150;
// `gaussian_ll` blvm/utils/log_likelihoods.py:17-39 with epsilon = 0): a frame's 2 activations -> Linear(2->2) -> (mu, raw sd),
// sd = softplus_beta(raw) + eps, ll = -(y-mu)^2/(2 sd^2) - log sd - log(2 pi)/2; masked per-utterance float64 sums.
// One lane per frame, 8-byte coalesced accesses; frame -> (utterance, sample) map as in dmol_kernel.
template <bool BWD>
__global__ __launch_bounds__(256) void gauss_head_kernel(DmolArgs a) {
  __shared__ double part[256];
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  const FrameCoord fc = frame_coord(a, f);
  const bool lds_path = !BWD && a.B <= 256;
  if (lds_path) part[threadIdx.x] = 0.0;
  float2 d = make_float2(0.f, 0.f);
  if (f < a.n_frames) d = reinterpret_cast<const float2*>(a.dec)[f];
  float w00 = 1.f, w01 = 0.f, w10 = 0.f, w11 = 1.f, b0 = 0.f, b1 = 0.f;
  if (a.W != nullptr) { w00 = a.W[0]; w01 = a.W[1]; w10 = a.W[2]; w11 = a.W[3]; b0 = a.bias[0]; b1 = a.bias[1]; }
  const float mu = fmaf(w00, d.x, fmaf(w01, d.y, b0)), raw = fmaf(w10, d.x, fmaf(w11, d.y, b1));
  const float sd = softplus_beta(raw, a.sd_beta, 1.f / a.sd_beta) + a.sd_eps;
  const float yv = fc.valid ? a.y[(size_t)fc.b * a.T + fc.tau] : 0.f;
  const float isd = 1.f / sd, z = (yv - mu) * isd;
  if (!BWD) {
    const float ll = fc.valid ? -0.5f * z * z - logf(sd) - 0.91893853320467274f : 0.f;
    if (a.ll_twise != nullptr && fc.valid) a.ll_twise[(size_t)fc.b * a.T + fc.tau] = ll;
    if (lds_path) {
      __syncthreads();
      if (fc.valid) atomicAdd(&part[fc.b], (double)ll);
      __syncthreads();
      if ((int)threadIdx.x < a.B && part[threadIdx.x] != 0.0) atomicAdd(a.log_prob + threadIdx.x, part[threadIdx.x]);
    } else if (fc.valid) {
      atomicAdd(a.log_prob + fc.b, (double)ll);
    }
  } else {
    if (f >= a.n_frames) return;
    const float g = fc.valid ? a.g_b[fc.b] : 0.f;
    const float dmu = g * z * isd;
    const float draw = g * (z * z - 1.f) * isd * sigmoidf_(a.sd_beta * raw);
    if (a.d_par != nullptr) reinterpret_cast<float2*>(a.d_par)[f] = make_float2(dmu, draw);
    reinterpret_cast<float2*>(a.d_dec)[f] = make_float2(fmaf(w00, dmu, w10 * draw), fmaf(w01, dmu, w11 * draw));
  }
}

template <bool BWD>
int launch_gauss(const DmolArgs& a, hipStream_t s) {
  const long long blocks = (a.n_frames + 255) / 256;
  BLVM_REQUIRE(blocks < (1ll << 31), "gauss_head: too many frames");
  hipLaunchKernelGGL((gauss_head_kernel<BWD>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return BLVM_OK;
}

}  // namespace
}  // namespace blvm

namespace blvm {
namespace {

// ---- samplers / modes of the mixture heads ---------------------------------------------------------------------------
// par [n, 30] = head outputs (logits | locations | raw scales).  u [n, 10] in (0,1): Gumbel-max component pick
// (`rsample_discretized_logistic_mixture` blvm/utils/variational.py:333-345, `rsample_gaussian_mixture` :179-191); NULL: the
// arg-max-logit component (`mode`, distributions.py:359-368).  v [n]: the component's own noise — kind 0: uniform in (0,1),
// x = loc + exp(max(raw, log_eps)) * (log v - log(1-v)), clamped to [-1,1] (:283-305); kind 1: standard normal,
// x = mu + (softplus_beta(raw) + eps) * v (sd_beta == 0: the scales are already standard deviations).  NULL: x = the location.
__global__ __launch_bounds__(256) void mix_sample_kernel(const float* __restrict__ par, const float* __restrict__ u,
                                                         const float* __restrict__ v, long long n, int kind, float log_eps,
                                                         float sd_beta, float sd_eps, float* __restrict__ out) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const float* p = par + f * F_MAX;
  int best = 0;
  float bv = -INFINITY;
#pragma unroll
  for (int m = 0; m < NMIX; ++m) {
    float s = p[m];
    if (u != nullptr) s -= logf(-logf(u[f * NMIX + m]));
    if (s > bv) { bv = s; best = m; }  // first maximum, as torch.argmax
  }
  const float loc = p[NMIX + best], raw = p[2 * NMIX + best];
  float x = loc;
  if (v != nullptr) {
    const float vv = v[f];
    if (kind == 0) {
      x = loc + expf(fmaxf(raw, log_eps)) * (logf(vv) - logf(1.f - vv));
      x = fminf(fmaxf(x, -1.f), 1.f);
    } else {
      x = loc + (sd_beta > 0.f ? softplus_beta(raw, sd_beta, 1.f / sd_beta) + sd_eps : raw) * vv;  // sd_beta == 0: raw IS the sd
    }
  }
  out[f] = x;
}

}  // namespace
}  // namespace blvm

extern "C" int blvm_mix_sample(const float* par, const float* u, const float* v, long long n, int num_mix, int kind,
                               float log_eps, float sd_beta, float sd_eps, float* out, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(par && out && n >= 0 && num_mix == NMIX && (kind == 0 || kind == 1), "mix_sample: bad arguments");
  if (n == 0) return BLVM_OK;
  const long long blocks = (n + 255) / 256;
  BLVM_REQUIRE(blocks < (1ll << 31), "mix_sample: too many frames");
  hipLaunchKernelGGL(mix_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), par, u, v, n, kind,
                     log_eps, sd_beta, sd_eps, out);
  BLVM_CHECK_LAUNCH("mix_sample");
  return BLVM_OK;
}

extern "C" int blvm_gauss_head_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                                   const int32_t* x_sl, int B, int T, int Tp, int S, float sd_beta, float sd_eps,
                                   double* log_prob, float* ll_twise, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, NMIX, 2, layout);
  if (rc) return rc;
  BLVM_REQUIRE(log_prob != nullptr && sd_beta > 0.f && aligned16(dec), "gauss_head_fwd: bad arguments");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, 2, 0.f);
  a.kind = 2; a.sd_beta = sd_beta; a.sd_eps = sd_eps;
  a.log_prob = log_prob;
  a.ll_twise = ll_twise;
  rc = launch_gauss<false>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("gauss_head_fwd");
  return BLVM_OK;
}

extern "C" int blvm_gauss_head_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                                   const int32_t* x_sl, const float* g_b, int B, int T, int Tp, int S, float sd_beta,
                                   float sd_eps, float* d_dec, float* d_par, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, NMIX, 2, layout);
  if (rc) return rc;
  BLVM_REQUIRE(g_b && d_dec && (d_par || !W) && sd_beta > 0.f && aligned16(dec) && aligned16(d_dec), "gauss_head_bwd: bad arguments");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, 2, 0.f);
  a.kind = 2; a.sd_beta = sd_beta; a.sd_eps = sd_eps;
  a.g_b = g_b;
  a.d_dec = d_dec;
  a.d_par = d_par;
  rc = launch_gauss<true>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("gauss_head_bwd");
  return BLVM_OK;
}

extern "C" int blvm_gmm_fwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                            const int32_t* x_sl, int B, int T, int Tp, int S, int num_mix, float sd_beta, float sd_eps,
                            double* log_prob, float* ll_twise, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, num_mix, 2, layout);
  if (rc) return rc;
  BLVM_REQUIRE(log_prob != nullptr && sd_beta > 0.f, "gmm_fwd: bad arguments");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, 2, 0.f);
  a.kind = 1; a.sd_beta = sd_beta; a.sd_eps = sd_eps;
  a.log_prob = log_prob;
  a.ll_twise = ll_twise;
  rc = launch_dmol<false>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("gmm_fwd");
  return BLVM_OK;
}

extern "C" int blvm_gmm_bwd(const float* dec, int layout, const float* W, const float* bias, const float* y,
                            const int32_t* x_sl, const float* g_b, int B, int T, int Tp, int S, int num_mix, float sd_beta,
                            float sd_eps, float* d_dec, float* d_par, void* stream) {
  using namespace blvm;
  int rc = check_common(dec, W, bias, y, x_sl, B, T, Tp, S, num_mix, 2, layout);
  if (rc) return rc;
  BLVM_REQUIRE(g_b && d_dec && (d_par || !W) && sd_beta > 0.f, "gmm_bwd: bad arguments");
  DmolArgs a = make_args(dec, layout, W, bias, y, x_sl, B, T, Tp, S, 2, 0.f);
  a.kind = 1; a.sd_beta = sd_beta; a.sd_eps = sd_eps;
  a.g_b = g_b;
  a.d_dec = d_dec;
  a.d_par = d_par;
  rc = launch_dmol<true>(a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("gmm_bwd");
  return BLVM_OK;
}
