// wavenet.hip — K10: dilated causal convolution (kernel size 2) and the gated residual WaveNet block, forward + backward.
//
// Replaces CausalConv1d (blvm/models/wavenet/wavenet_modules.py:14-50) and Conv1dResidualGLU (:53-117):
//   pre = Conv1d(C -> 2C, k=2, dilation=d)(x);  act = tanh(pre[:C]) * sigmoid(pre[C:])   (GatedTanhUnit, activations.py:5-13)
//   rs  = Conv1d(C -> C+S, k=1)(act);  o = (rs[:C] + x[..., d:]) * sqrt(0.5);  skip += rs[C:][..., -T:]
// Layout: TIME-MAJOR channel-last [L, B, C].  A dilation then is a plain row offset of d*B rows, "the last T frames"
// and "what the dilated kernel ate" are row suffixes, and every convolution is a dense [rows, C] x [C, N] product:
// the two taps of the k=2 kernel are two MFMA GEMMs over shifted views of the same buffer (gemm.hip, second one
// accumulating), the 1x1 convolution is one GEMM.  The element-wise pieces (gate, residual/skip, their derivatives)
// are 16-byte-vectorised streaming kernels.  Nothing is padded inside the stack: like the reference, block i
// consumes d_i frames on the left.
#include <vector>

#include "common.h"

namespace blvm {
namespace {

inline int pick_split(int M, int N, int K) { return gemm_pick_split(M, N, K); }

inline dim3 ew_grid(size_t n_items) {
  size_t blocks = (n_items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// W [N,K,2] (Conv1d weight, taps interleaved) -> W0 [N,K] (tap on x[t]) and W1 [N,K] (tap on x[t+d])
__global__ __launch_bounds__(256) void split_taps_kernel(const float* __restrict__ W, float* __restrict__ W0,
                                                         float* __restrict__ W1, size_t nk) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += (size_t)gridDim.x * 256) {
    const float2 w = reinterpret_cast<const float2*>(W)[i];
    W0[i] = w.x;
    W1[i] = w.y;
  }
}

// dW [N,K,2] += (dW0, dW1)
__global__ __launch_bounds__(256) void merge_taps_kernel(const float* __restrict__ dW0, const float* __restrict__ dW1,
                                                         float* __restrict__ dW, size_t nk) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += (size_t)gridDim.x * 256) {
    float2 w = reinterpret_cast<float2*>(dW)[i];
    w.x += dW0[i];
    w.y += dW1[i];
    reinterpret_cast<float2*>(dW)[i] = w;
  }
}

// act[r,c] = tanh(pre[r,c]) * sigmoid(pre[r,C+c]);  one thread per 4 channels
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ pre, float* __restrict__ act, size_t rows, int C) {
  const int c4n = C / 4;
  const size_t n = rows * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 a = *reinterpret_cast<const float4*>(pre + r * 2 * C + c);
    const float4 b = *reinterpret_cast<const float4*>(pre + r * 2 * C + C + c);
    float4 o;
    o.x = tanhf(a.x) * sigmoidf_(b.x); o.y = tanhf(a.y) * sigmoidf_(b.y);
    o.z = tanhf(a.z) * sigmoidf_(b.z); o.w = tanhf(a.w) * sigmoidf_(b.w);
    *reinterpret_cast<float4*>(act + r * C + c) = o;
  }
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ d_act,
                                                       float* __restrict__ d_pre, size_t rows, int C) {
  const int c4n = C / 4;
  const size_t n = rows * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 a = *reinterpret_cast<const float4*>(pre + r * 2 * C + c);
    const float4 b = *reinterpret_cast<const float4*>(pre + r * 2 * C + C + c);
    const float4 g = *reinterpret_cast<const float4*>(d_act + r * C + c);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, gv[4] = {g.x, g.y, g.z, g.w};
    float da[4], db[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ta = tanhf(av[k]), sb = sigmoidf_(bv[k]);
      da[k] = gv[k] * sb * (1.f - ta * ta);
      db[k] = gv[k] * ta * sb * (1.f - sb);
    }
    *reinterpret_cast<float4*>(d_pre + r * 2 * C + c) = make_float4(da[0], da[1], da[2], da[3]);
    *reinterpret_cast<float4*>(d_pre + r * 2 * C + C + c) = make_float4(db[0], db[1], db[2], db[3]);
  }
}

// o = (rs[:, :C] + xres) * inv_std;  skip[r - off, :] += rs[r, C:] for rows r >= off   (one thread per element)
__global__ __launch_bounds__(256) void resskip_fwd_kernel(const float* __restrict__ rs, const float* __restrict__ xres,
                                                          float* __restrict__ o, float* __restrict__ skip, size_t rows,
                                                          size_t off, int C, int S, float inv_std) {
  const int W = C + S;
  const size_t n = rows * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / W;
    const int c = (int)(i - r * W);
    const float v = rs[i];
    if (c < C) {
      if (o != nullptr) o[r * C + c] = (v + xres[r * C + c]) * inv_std;
    } else if (r >= off) {
      skip[(r - off) * S + (c - C)] += v;
    }
  }
}

// d_rs[r, :C] = d_o * inv_std (0 if no d_o);  d_rs[r, C:] = d_skip[r - off] for r >= off else 0;  d_xres = d_o * inv_std
__global__ __launch_bounds__(256) void resskip_bwd_kernel(const float* __restrict__ d_o, const float* __restrict__ d_skip,
                                                          float* __restrict__ d_rs, float* __restrict__ d_xres, size_t rows,
                                                          size_t off, int C, int S, float inv_std) {
  const int W = C + S;
  const size_t n = rows * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / W;
    const int c = (int)(i - r * W);
    if (c < C) {
      const float g = d_o != nullptr ? d_o[r * C + c] * inv_std : 0.f;
      d_rs[i] = g;
      d_xres[r * C + c] = g;
    } else {
      d_rs[i] = r >= off ? d_skip[(r - off) * S + (c - C)] : 0.f;
    }
  }
}

// y = act(x * scale), act = ReLU / LeakyReLU(slope)
__global__ __launch_bounds__(256) void scale_act_kernel(const float* __restrict__ x, float scale, float slope,
                                                        float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i] * scale;
    y[i] = v > 0.f ? v : v * slope;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused forward of one gated residual block (C in {32, 64, 96}, 16 <= S <= C).  The unfused sequence (two tap GEMMs, gate
// pass, 1x1 GEMM, residual/skip pass) moves ~20 floats per row and channel through HBM; everything between the block's
// input and its outputs fits a wave's LDS slice, so here a wave owns 16 rows end to end:
//   x[r], x[r + d B] -> LDS;  pre = X0 W0^T + X1 W1^T + b (tanh tile and sigmoid tile of the same channels side by side on
//   the matrix pipe);  act = tanh * sigma in registers -> LDS;  rs = act Wrs^T + b;  o = (rs[:C] + X1) * inv_std;
//   skip += rs[C:];  pre / act are written once for backward (16 rows of a [rows, 2C] buffer are one contiguous run).
// Weights arrive in the T16 operand layout (one contiguous 1 KB read per fragment, common.h); the four waves of a workgroup
// walk the column tiles in step (one barrier per tile) so three of every four fragment reads hit the CU's L1.
// ---------------------------------------------------------------------------------------------------------------
struct FusedFwdArgs {
  const float* x;                      // [L_in*B, C]
  const float *W0, *W1, *Wrs;          // T16: [2C,C], [2C,C], [C+S,C]
  const float *conv_b, *rs_b;          // [2C], [C+S]
  float *pre, *act, *o, *skip;         // [rows,2C], [rows,C], [rows,C] or null, [T_skip*B,S]
  size_t rows, shift, off;             // off: first row that contributes to the skip sum
  float inv_std;
};

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Two interleaved 16x16 products over the 4 k-values a lane holds of each operand (w = weights as the A operand, x = data as the B
// operand).  fp32: four v_mfma_f32_16x16x4_f32 per product, the two accumulator chains alternating; BF (operand_bf16(), the
// reference's --use_amp regime, experiments/experiment_wavenet_audio.py): the same 4 k-values rounded to bf16 are exactly one
// v_mfma_f32_16x16x16_bf16 per product — a quarter of the matrix-pipe time of kernels whose fp32 form is bound by it.
typedef short wn_s16x4 __attribute__((ext_vector_type(4)));
typedef float wn_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 wn_b2 __attribute__((ext_vector_type(2)));
typedef unsigned wn_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wn_s16x4 pk4(float a, float b, float c, float d) {
  const wn_u2 q = {__builtin_bit_cast(unsigned, __builtin_convertvector((wn_f2){a, b}, wn_b2)), __builtin_bit_cast(unsigned, __builtin_convertvector((wn_f2){c, d}, wn_b2))};
  return __builtin_bit_cast(wn_s16x4, q);
}
__device__ __forceinline__ wn_s16x4 pk4(const float4& v) { return pk4(v.x, v.y, v.z, v.w); }
template <bool BF>
__device__ __forceinline__ void mma4x2(f32x4& a0, const float4& w0, const float4& x0, f32x4& a1, const float4& w1, const float4& x1) {
  if constexpr (BF) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pk4(w0), pk4(x0), a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pk4(w1), pk4(x1), a1, 0, 0, 0);
  } else {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, a1, 0, 0, 0);
  }
}

template <int C, int S, bool BF>
__global__ __launch_bounds__(256, 2) void wn_block_fused_fwd_kernel(FusedFwdArgs a) {
  constexpr int LDX = C + 4, KC = C / 16, NT = C / 16, NR = (C + S) / 16;
  constexpr int WAVE_FLOATS = 16 * 3 * LDX;
  constexpr int NX = (16 * C / 4 + 63) / 64;  // 16-byte pieces of a 16-row slab per lane
  extern __shared__ __align__(16) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, q = lane >> 4, cc = lane & 15;
  float* sX0 = smem + wave * WAVE_FLOATS;
  float* sX1 = sX0 + 16 * LDX;
  float* sAct = sX1 + 16 * LDX;
  const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 16;
#ifdef BLVM_WN_PROF
  long long tk = __builtin_readcyclecounter(), ph[6] = {0, 0, 0, 0, 0, 0};
#define WN_TICK(k) { const long long n__ = __builtin_readcyclecounter(); ph[k] += n__ - tk; tk = n__; }
#else
#define WN_TICK(k)
#endif

  // a "stage" of the convolution = JH k-chunks of one (tanh, sigmoid) tile pair: its 4 JH weight fragments are what one
  // register set holds (C = 96: half a pair, so that two sets + the 1x1 sets stay under 256 VGPRs at two waves per SIMD)
  constexpr int JH = KC > 4 ? KC / 2 : KC, SPC = KC / JH, NST = NT * SPC;
  struct ConvFrag { float4 t0[JH], s0[JH], t1[JH], s1[JH], bT, bS; };
  struct RsFrag { float4 f0[KC], f1[KC], b0, b1; };
  auto load_conv = [&](ConvFrag& F, int st) {
    const int ct = st / SPC, j0 = (st - ct * SPC) * JH;
    const float* wT0 = a.W0 + (size_t)(ct * 16) * C + 4 * lane + 256 * j0;
    const float* wS0 = a.W0 + (size_t)((NT + ct) * 16) * C + 4 * lane + 256 * j0;
    const float* wT1 = a.W1 + (size_t)(ct * 16) * C + 4 * lane + 256 * j0;
    const float* wS1 = a.W1 + (size_t)((NT + ct) * 16) * C + 4 * lane + 256 * j0;
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      F.t0[j] = *reinterpret_cast<const float4*>(wT0 + 256 * j);
      F.s0[j] = *reinterpret_cast<const float4*>(wS0 + 256 * j);
      F.t1[j] = *reinterpret_cast<const float4*>(wT1 + 256 * j);
      F.s1[j] = *reinterpret_cast<const float4*>(wS1 + 256 * j);
    }
    F.bT = *reinterpret_cast<const float4*>(a.conv_b + ct * 16 + 4 * q);
    F.bS = *reinterpret_cast<const float4*>(a.conv_b + C + ct * 16 + 4 * q);
  };
  auto load_rs = [&](RsFrag& G, int ct) {  // column tiles ct, ct + 1 (NR is even)
    const float* w0 = a.Wrs + (size_t)(ct * 16) * C + 4 * lane;
    const float* w1 = a.Wrs + (size_t)((ct + 1) * 16) * C + 4 * lane;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      G.f0[j] = *reinterpret_cast<const float4*>(w0 + 256 * j);
      G.f1[j] = *reinterpret_cast<const float4*>(w1 + 256 * j);
    }
    G.b0 = *reinterpret_cast<const float4*>(a.rs_b + ct * 16 + 4 * q);
    G.b1 = *reinterpret_cast<const float4*>(a.rs_b + (ct + 1) * 16 + 4 * q);
  };
  ConvFrag F0, F1;
  RsFrag G0, G1;
  static_assert(NST % 2 == 0 && NR % 4 == 0 && KC % JH == 0, "fused WaveNet block: tile counts");

  // 1. the two taps' input rows (16 consecutive rows = one contiguous run of 16 C floats each): every load is issued before
  // the first LDS write, and the first pair's weight fragments travel with them
  {
    float4 v0[NX], v1[NX];
#pragma unroll
    for (int n = 0; n < NX; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / C, col = (i * 4) - row * C;
      const size_t gr = r0 + row;
      v0[n] = v1[n] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 16 * C / 4 && gr < a.rows) {
        v0[n] = *reinterpret_cast<const float4*>(a.x + gr * C + col);
        v1[n] = *reinterpret_cast<const float4*>(a.x + (gr + a.shift) * C + col);
      }
    }
    load_conv(F0, 0);
#pragma unroll
    for (int n = 0; n < NX; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / C, col = (i * 4) - row * C;
      if (i < 16 * C / 4) {
        *reinterpret_cast<float4*>(sX0 + row * LDX + col) = v0[n];
        *reinterpret_cast<float4*>(sX1 + row * LDX + col) = v1[n];
      }
    }
  }
  wave_lds_fence();
  WN_TICK(0)

  // 2. dilated convolution + gate, one (tanh, sigmoid) pair of column tiles at a time.  The weight fragments of the NEXT pair
  // are loaded into a second register set while this pair runs on the matrix pipe (the two sets swap roles by position in a
  // 2-pair loop body); the barrier only keeps the four waves on the same fragments (L1).  Two workgroups share a CU, so one
  // wave's epilogue (tanh, sigma, stores) runs under the other's MFMAs.  pre leaves straight from the accumulators (D layout:
  // column lane & 15, rows 4 (lane >> 4) + r — 64-byte runs that pair up into full lines in L2).
  const float* ap0 = sX0 + rr * LDX + 4 * q;
  const float* ap1 = sX1 + rr * LDX + 4 * q;
  f32x4 aT = {0.f, 0.f, 0.f, 0.f}, aS = {0.f, 0.f, 0.f, 0.f};
  auto conv_stage = [&](const ConvFrag& F, int st) {
    const int ct = st / SPC, j0 = (st - ct * SPC) * JH;
    if (j0 == 0) { aT = (f32x4){0.f, 0.f, 0.f, 0.f}; aS = aT; }
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const float4 x0 = *reinterpret_cast<const float4*>(ap0 + 16 * (j0 + j));
      const float4 x1 = *reinterpret_cast<const float4*>(ap1 + 16 * (j0 + j));
      mma4x2<BF>(aT, F.t0[j], x0, aS, F.s0[j], x0);
      mma4x2<BF>(aT, F.t1[j], x1, aS, F.s1[j], x1);
    }
    if (j0 + JH < KC) return;
    // weights as the A operand: the accumulator holds data row lane & 15, channels 4 (lane >> 4) + r — 16-byte pieces of a row
    const int col = ct * 16 + 4 * q;
    const float4 pT = make_float4(aT[0] + F.bT.x, aT[1] + F.bT.y, aT[2] + F.bT.z, aT[3] + F.bT.w);
    const float4 pS = make_float4(aS[0] + F.bS.x, aS[1] + F.bS.y, aS[2] + F.bS.z, aS[3] + F.bS.w);
    *reinterpret_cast<float4*>(sAct + rr * LDX + col) = make_float4(tanhf(pT.x) * sigmoidf_(pS.x), tanhf(pT.y) * sigmoidf_(pS.y),
                                                                    tanhf(pT.z) * sigmoidf_(pS.z), tanhf(pT.w) * sigmoidf_(pS.w));
    if (r0 + rr < a.rows) {
      *reinterpret_cast<float4*>(a.pre + (r0 + rr) * 2 * C + col) = pT;
      *reinterpret_cast<float4*>(a.pre + (r0 + rr) * 2 * C + C + col) = pS;
    }
  };
  for (int st = 0; st < NST; st += 2) {
    asm volatile("s_barrier" ::: "memory");
    load_conv(F1, st + 1);
    conv_stage(F0, st);
    if (st + 2 < NST) load_conv(F0, st + 2);
    else load_rs(G0, 0);
    conv_stage(F1, st + 1);
  }
  wave_lds_fence();
  WN_TICK(1)

  // 3. the gate output for backward (a contiguous run)
  for (int i = lane; i < 16 * C / 4; i += 64) {
    const int row = (i * 4) / C, col = (i * 4) - row * C;
    if (r0 + row < a.rows) *reinterpret_cast<float4*>(a.act + (r0 + row) * C + col) = *reinterpret_cast<const float4*>(sAct + row * LDX + col);
  }
  WN_TICK(2)

  // 4. 1x1 convolution (residual | skip columns), two column tiles at a time; the residual columns take x[r + d B]; the
  // skip rows are accumulated in place (the stack's blocks run one after the other, each the only writer of its launch),
  // their old values requested before the MFMAs
  const float* ap = sAct + rr * LDX + 4 * q;
  auto rs_pair = [&](const RsFrag& G, int ct) {
    const size_t gr = r0 + rr;
    const bool live = gr < a.rows;
    float4 old[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = (ct + h) * 16 + 4 * q;
      old[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col >= C && live && gr >= a.off) old[h] = *reinterpret_cast<const float4*>(a.skip + (gr - a.off) * S + (col - C));
    }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(ap + 16 * j);
      mma4x2<BF>(c0, G.f0[j], v, c1, G.f1[j], v);
    }
    if (!live) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = (ct + h) * 16 + 4 * q;
      const float4 b = h ? G.b1 : G.b0;
      const f32x4 c = h ? c1 : c0;
      float4 v = make_float4(c[0] + b.x, c[1] + b.y, c[2] + b.z, c[3] + b.w);
      if (col < C) {
        if (a.o != nullptr) {
          const float4 x1 = *reinterpret_cast<const float4*>(sX1 + rr * LDX + col);
          *reinterpret_cast<float4*>(a.o + gr * C + col) =
              make_float4((v.x + x1.x) * a.inv_std, (v.y + x1.y) * a.inv_std, (v.z + x1.z) * a.inv_std, (v.w + x1.w) * a.inv_std);
        }
      } else if (gr >= a.off) {
        *reinterpret_cast<float4*>(a.skip + (gr - a.off) * S + (col - C)) = make_float4(old[h].x + v.x, old[h].y + v.y, old[h].z + v.z, old[h].w + v.w);
      }
    }
  };
  for (int ct = 0; ct < NR; ct += 4) {
    asm volatile("s_barrier" ::: "memory");
    load_rs(G1, ct + 2);
    rs_pair(G0, ct);
    asm volatile("s_barrier" ::: "memory");
    if (ct + 4 < NR) load_rs(G0, ct + 4);
    rs_pair(G1, ct + 2);
  }
  WN_TICK(3)
#ifdef BLVM_WN_PROF
  if (blockIdx.x == 1000 && lane == 0) for (int k = 0; k < 4; ++k) a.act[k + 8 * wave] = (float)ph[k];
#endif
}

template <int C, int S>
int launch_fused_fwd(const FusedFwdArgs& a, hipStream_t s) {
  constexpr size_t lds = sizeof(float) * 4 * 16 * 3 * (C + 4);
  static_assert(2 * lds <= 160 * 1024, "fused WaveNet block: two workgroups per CU");
  auto kern = operand_bf16() ? wn_block_fused_fwd_kernel<C, S, true> : wn_block_fused_fwd_kernel<C, S, false>;
  if (lds > 64 * 1024) BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t tiles = (a.rows + 63) / 64;
  BLVM_REQUIRE(tiles < (1ull << 31), "wavenet_block_fwd: too many rows");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, s, a);
  return BLVM_OK;
}

// env BLVM_WN_FUSED=0 keeps the unfused sequence (experiments)
inline bool fused_fwd_enabled() {
  static int v = [] {
    const char* e = getenv("BLVM_WN_FUSED");
    return e ? atoi(e) : 1;
  }();
  return v != 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused backward of the block's data path, two kernels with the forward kernel's structure (a wave owns 16 rows, weights in
// T16, the next tile's fragments loaded under the current tile's MFMAs):
//   A: d_rs = [d_o * inv_std | d_skip] (written for the weight-gradient GEMM / column sum), d_act = d_rs Wrs, through the gate
//      derivative to d_pre (written once; pre is read in the accumulators' layout).
//   B: d_x[r] = d_pre[r] W0 + d_pre[r - d B] W1 + d_o[r - d B] * inv_std in ONE pass over d_x (the unfused sequence zero-fills
//      d_x and accumulates the two taps into it with two read-modify-write GEMMs).
// ---------------------------------------------------------------------------------------------------------------
struct FusedBwdAArgs {
  const float *d_o, *d_skip, *pre, *WrsT;  // WrsT: T16 [C, C+S]
  float *d_rs, *d_pre;                     // [rows, C+S], [rows, 2C]
  size_t rows, off;
  float inv_std;
};

template <int C, int S, bool BF>
__global__ __launch_bounds__(256, 2) void wn_block_fused_bwd_a_kernel(FusedBwdAArgs a) {
  constexpr int W = C + S, LDR = W + 4, KR = W / 16, NT = C / 16, KH = KR / 2;
  constexpr int NP = (16 * W / 4 + 63) / 64;
  static_assert(NT % 2 == 0 && KR % 2 == 0, "fused WaveNet block: tile counts");
  extern __shared__ __align__(16) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, q = lane >> 4, cc = lane & 15;
  float* sD = smem + wave * 16 * LDR;
  const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 16;
  struct Frag { float4 f[KR], pT, pS; };
  auto load_tile = [&](Frag& F, int ct) {
    const float* w = a.WrsT + (size_t)(ct * 16) * W + 4 * lane;
#pragma unroll
    for (int j = 0; j < KR; ++j) F.f[j] = *reinterpret_cast<const float4*>(w + 256 * j);
    const size_t o = (r0 + rr < a.rows ? r0 + rr : 0) * 2 * C + ct * 16 + 4 * q;
    F.pT = *reinterpret_cast<const float4*>(a.pre + o);
    F.pS = *reinterpret_cast<const float4*>(a.pre + o + C);
  };
  Frag F0, F1;
  // 1. d_rs rows: [d_o * inv_std | d_skip of the rows that fed the skip sum]
  {
    float4 v[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / W, col = (i * 4) - row * W;
      const size_t gr = r0 + row;
      v[n] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 16 * W / 4 && gr < a.rows) {
        if (col < C) {
          if (a.d_o != nullptr) {
            v[n] = *reinterpret_cast<const float4*>(a.d_o + gr * C + col);
            v[n].x *= a.inv_std; v[n].y *= a.inv_std; v[n].z *= a.inv_std; v[n].w *= a.inv_std;
          }
        } else if (gr >= a.off) {
          v[n] = *reinterpret_cast<const float4*>(a.d_skip + (gr - a.off) * S + (col - C));
        }
      }
    }
    load_tile(F0, 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / W, col = (i * 4) - row * W;
      if (i < 16 * W / 4) {
        *reinterpret_cast<float4*>(sD + row * LDR + col) = v[n];
        if (r0 + row < a.rows) *reinterpret_cast<float4*>(a.d_rs + (r0 + row) * W + col) = v[n];
      }
    }
  }
  wave_lds_fence();
  // 2. d_act = d_rs Wrs per column tile (two half-K chains), gate derivative in the accumulators' layout
  const float* ap = sD + rr * LDR + 4 * q;
  auto tile = [&](const Frag& F, int ct) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KH; ++j) {
      const float4 u = *reinterpret_cast<const float4*>(ap + 16 * j);
      const float4 v = *reinterpret_cast<const float4*>(ap + 16 * (KH + j));
      mma4x2<BF>(c0, F.f[j], u, c1, F.f[KH + j], v);
    }
    // accumulators: data row lane & 15, channels 4 (lane >> 4) + r
    const size_t gr = r0 + rr;
    if (gr >= a.rows) return;
    const float pt[4] = {F.pT.x, F.pT.y, F.pT.z, F.pT.w}, ps[4] = {F.pS.x, F.pS.y, F.pS.z, F.pS.w};
    float dt[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float g = c0[r] + c1[r];
      const float ta = tanhf(pt[r]), sb = sigmoidf_(ps[r]);
      dt[r] = g * sb * (1.f - ta * ta);
      ds[r] = g * ta * sb * (1.f - sb);
    }
    const size_t o = gr * 2 * C + ct * 16 + 4 * q;
    *reinterpret_cast<float4*>(a.d_pre + o) = make_float4(dt[0], dt[1], dt[2], dt[3]);
    *reinterpret_cast<float4*>(a.d_pre + o + C) = make_float4(ds[0], ds[1], ds[2], ds[3]);
  };
  for (int ct = 0; ct < NT; ct += 2) {
    asm volatile("s_barrier" ::: "memory");
    load_tile(F1, ct + 1);
    tile(F0, ct);
    if (ct + 2 < NT) load_tile(F0, ct + 2);
    tile(F1, ct + 1);
  }
}

struct FusedBwdBArgs {
  const float *d_pre, *d_o, *W0T, *W1T;  // W0T / W1T: T16 [C, 2C] (the taps' weights transposed)
  float* d_x;                            // [rows + shift, C]
  size_t rows, shift;
  float inv_std;
};

template <int C, bool BF>
__global__ __launch_bounds__(256, 2) void wn_block_fused_bwd_b_kernel(FusedBwdBArgs a) {
  constexpr int K2 = 2 * C, LDA = K2 + 4, KR = K2 / 16, NT = C / 16, KH = KR / 2;
  constexpr int NP = (16 * K2 / 4 + 63) / 64;
  static_assert(NT % 2 == 0 && KR % 2 == 0, "fused WaveNet block: tile counts");
  extern __shared__ __align__(16) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane & 15, q = lane >> 4, cc = lane & 15;
  float* sA = smem + wave * 16 * LDA;
  const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 16;  // rows of d_x
  const size_t rows_in = a.rows + a.shift;
  struct Frag { float4 f[KR]; };
  auto load_tile = [&](Frag& F, const float* WT, int ct) {
    const float* w = WT + (size_t)(ct * 16) * K2 + 4 * lane;
#pragma unroll
    for (int j = 0; j < KR; ++j) F.f[j] = *reinterpret_cast<const float4*>(w + 256 * j);
  };
  // d_pre rows [first, first + 16) -> LDS (rows outside [0, rows) are zero); `first` may be "negative" (tap 1 near the start)
  auto stage = [&](long long first) {
    float4 v[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / K2, col = (i * 4) - row * K2;
      const long long gr = first + row;
      v[n] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 16 * K2 / 4 && gr >= 0 && gr < (long long)a.rows) v[n] = *reinterpret_cast<const float4*>(a.d_pre + (size_t)gr * K2 + col);
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const int i = lane + 64 * n;
      const int row = (i * 4) / K2, col = (i * 4) - row * K2;
      if (i < 16 * K2 / 4) *reinterpret_cast<float4*>(sA + row * LDA + col) = v[n];
    }
  };
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* ap = sA + rr * LDA + 4 * q;
  auto tile = [&](const Frag& F, f32x4& c) {
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KH; ++j) {
      const float4 u = *reinterpret_cast<const float4*>(ap + 16 * j);
      const float4 v = *reinterpret_cast<const float4*>(ap + 16 * (KH + j));
      mma4x2<BF>(c, F.f[j], u, c1, F.f[KH + j], v);
    }
    c += c1;
  };
  Frag F0, F1;
#pragma unroll
  for (int tap = 0; tap < 2; ++tap) {
    const float* WT = tap ? a.W1T : a.W0T;
    load_tile(F0, WT, 0);
    stage(tap ? (long long)r0 - (long long)a.shift : (long long)r0);
    wave_lds_fence();
#pragma unroll
    for (int ct = 0; ct < NT; ct += 2) {
      asm volatile("s_barrier" ::: "memory");
      load_tile(F1, WT, ct + 1);
      tile(F0, acc[ct]);
      if (ct + 2 < NT) load_tile(F0, WT, ct + 2);
      tile(F1, acc[ct + 1]);
    }
    wave_lds_fence();  // every fragment read of this tap's rows is done before the buffer is refilled
  }
  const size_t gr = r0 + rr;  // accumulators: data row lane & 15, channels 4 (lane >> 4) + r
  if (gr < rows_in) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int col = ct * 16 + 4 * q;
      float4 v = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
      if (a.d_o != nullptr && gr >= a.shift) {
        const float4 g = *reinterpret_cast<const float4*>(a.d_o + (gr - a.shift) * C + col);
        v.x += g.x * a.inv_std; v.y += g.y * a.inv_std; v.z += g.z * a.inv_std; v.w += g.w * a.inv_std;
      }
      *reinterpret_cast<float4*>(a.d_x + gr * C + col) = v;
    }
  }
}

template <int C, int S>
int launch_fused_bwd(const FusedBwdAArgs& aa, const FusedBwdBArgs& ab, hipStream_t s) {
  constexpr size_t lds_a = sizeof(float) * 4 * 16 * (C + S + 4), lds_b = sizeof(float) * 4 * 16 * (2 * C + 4);
  static_assert(2 * lds_a <= 160 * 1024 && 2 * lds_b <= 160 * 1024, "fused WaveNet block backward: two workgroups per CU");
  const size_t ta = (aa.rows + 63) / 64, tb = (ab.rows + ab.shift + 63) / 64;
  BLVM_REQUIRE(tb < (1ull << 31), "wavenet_block_bwd: too many rows");
  if (operand_bf16()) {
    hipLaunchKernelGGL((wn_block_fused_bwd_a_kernel<C, S, true>), dim3((unsigned)ta), dim3(256), lds_a, s, aa);
    hipLaunchKernelGGL((wn_block_fused_bwd_b_kernel<C, true>), dim3((unsigned)tb), dim3(256), lds_b, s, ab);
  } else {
    hipLaunchKernelGGL((wn_block_fused_bwd_a_kernel<C, S, false>), dim3((unsigned)ta), dim3(256), lds_a, s, aa);
    hipLaunchKernelGGL((wn_block_fused_bwd_b_kernel<C, false>), dim3((unsigned)tb), dim3(256), lds_b, s, ab);
  }
  return BLVM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight / bias gradients of the block: out[m][n] += sum_r A[r][m] B[r][n] with M = 2C, N = C small and ~10^6 rows.  A 64x64-tile
// split-K GEMM re-reads A once per column tile and B once per row tile (2.75 GB for the 1.18 GB of operands at C = 96); here
// a workgroup owns a slice of rows and the WHOLE [2C, C] output (every wave (2C/16)/4 row tiles x all column tiles, in
// accumulators), so the operands stream through LDS exactly once.  DUAL: both taps of the dilated convolution share the
// A = d_pre stream (B0 = x[r], B1 = x[r + d B]) and land interleaved in the Conv1d weight layout [2C, C, 2]; the column sums of
// A (the bias gradient) come from the same LDS tiles.
// ---------------------------------------------------------------------------------------------------------------
struct TsArgs {
  const float* A;    // [rows, 2C]
  const float* B0;   // [rows, C]
  const float* B1;   // [rows, C] (DUAL) or null
  size_t rows, chunks_per_wg;
  float* out;        // element (m, n) of tap t at out[(m * C + n) * ostride + t]
  int ostride;
  float* colsum;     // [2C] += column sums of A, or null
};

// NPW < C/16 (few rows: the reference's own batch of 4 utterances): a workgroup owns NPW column tiles of ONE tap instead of the whole
// output — NSPLIT = taps * (C/16) / NPW workgroups share a slice of rows (all on one XCD, so the slice's A rows come from HBM once)
// and there are NSPLIT times fewer row slices for the same number of workgroups, i.e. NSPLIT times fewer atomic adds on the same
// [2C, C(, 2)] words: with 512 slices of 8 chunks each the kernel spent 171 us in 18.9 M contended atomics for 30 us of MFMA.
template <int C, bool DUAL, bool BF, int NPW>
__global__ __launch_bounds__(256) void wn_ts_wgrad_kernel(TsArgs a) {
  constexpr bool SPLIT = NPW < C / 16;
  constexpr int M = 2 * C, N = C, MT = M / 16, MW = MT / 4;
  constexpr int KB = SPLIT ? 32 : 16;  // rows per staged chunk (few rows: the single prefetched chunk must cover an HBM round trip)
  constexpr int NT = SPLIT ? NPW : N / 16, NB = SPLIT ? 1 : (DUAL ? 2 : 1), NW_ = NT * 16;  // column tiles, taps and B columns of this workgroup
  constexpr int NSPLIT = SPLIT ? (DUAL ? 2 : 1) * (N / 16) / NPW : 1;
  static_assert(!SPLIT || (N / 16) % NPW == 0, "column tiles per workgroup must divide C/16");
  constexpr int LDA = M + 4, LDB = NW_ + 4;  // a lane's fragment = rows 4 kq .. 4 kq + 3 of one column: stride = 4 mod 8 spreads the 4 kq groups over all banks
  constexpr int NA4 = KB * M / 4, NB4 = KB * NW_ / 4;           // 16-byte pieces per chunk
  constexpr int PA = (NA4 + 255) / 256, PB = (NB4 + 255) / 256;
  __shared__ __align__(16) float sA[2][KB * LDA];
  __shared__ __align__(16) float sB[2][NB][KB * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
  // workgroup -> (row slice, part): ids are dealt round-robin over the 8 XCDs; the NSPLIT parts of a slice stay on one XCD
  const int q_ = blockIdx.x >> 3, part = SPLIT ? q_ % NSPLIT : 0;
  const size_t slice = SPLIT ? (size_t)(q_ / NSPLIT) * 8 + (blockIdx.x & 7) : blockIdx.x;
  const int tap = SPLIT ? part / ((N / 16) / NPW) : 0, j0 = SPLIT ? (part % ((N / 16) / NPW)) * NPW : 0;
  const float* Bsrc[2] = {SPLIT ? (tap ? a.B1 : a.B0) + j0 * 16 : a.B0, a.B1};
  const size_t c_begin = slice * a.chunks_per_wg;
  const size_t n_chunks = (a.rows + KB - 1) / KB;
  size_t c_end = c_begin + a.chunks_per_wg;
  if (c_end > n_chunks) c_end = n_chunks;
  if (c_begin >= c_end) return;

  f32x4 acc[NB][MW][NT];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  const bool do_csum = a.colsum != nullptr && part == 0 && tid < M;

  float4 ra[PA], rb[NB][PB];
  auto fetch = [&](size_t chunk) {
    const size_t row0 = chunk * KB;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int i = tid + 256 * p;
      const int row = (i * 4) / M, col = (i * 4) - row * M;
      ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NA4 && row0 + row < a.rows) ra[p] = *reinterpret_cast<const float4*>(a.A + (row0 + row) * M + col);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int i = tid + 256 * p;
      const int row = (i * 4) / NW_, col = (i * 4) - row * NW_;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        rb[b][p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < NB4 && row0 + row < a.rows) rb[b][p] = *reinterpret_cast<const float4*>(Bsrc[b] + (row0 + row) * N + col);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int i = tid + 256 * p;
      const int row = (i * 4) / M, col = (i * 4) - row * M;
      if (i < NA4) *reinterpret_cast<float4*>(&sA[buf][row * LDA + col]) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int i = tid + 256 * p;
      const int row = (i * 4) / NW_, col = (i * 4) - row * NW_;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (i < NB4) *reinterpret_cast<float4*>(&sB[buf][b][row * LDB + col]) = rb[b][p];
    }
  };
  fetch(c_begin);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (size_t c = c_begin; c < c_end; ++c) {
    const bool more = c + 1 < c_end;
    if (more) fetch(c + 1);  // in flight under this chunk's MFMAs
    const float* A_ = sA[buf];
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
      // fragments: k = rows 16 ks + 4 kq + e of the chunk, e = 0..3 (the same assignment for both operands), one MFMA per e
      float fa[MW][4], fb[NB][NT][4];
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const float* p = A_ + (16 * ks + 4 * kq) * LDA + (wave * MW + i) * 16 + li;
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[i][e] = p[e * LDA];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float* p = sB[buf][b] + (16 * ks + 4 * kq) * LDB + j * 16 + li;
#pragma unroll
          for (int e = 0; e < 4; ++e) fb[b][j][e] = p[e * LDB];
        }
      if constexpr (BF) {
        wn_s16x4 ha[MW], hb[NB][NT];
#pragma unroll
        for (int i = 0; i < MW; ++i) ha[i] = pk4(fa[i][0], fa[i][1], fa[i][2], fa[i][3]);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int j = 0; j < NT; ++j) hb[b][j] = pk4(fb[b][j][0], fb[b][j][1], fb[b][j][2], fb[b][j][3]);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[b][i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha[i], hb[b][j], acc[b][i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < MW; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[b][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[b][j][e], acc[b][i][j], 0, 0, 0);
      }
    }
    if (do_csum) {
#pragma unroll
      for (int r = 0; r < KB; ++r) csum += A_[r * LDA + tid];
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // D layout: column (n) lane & 15, rows (m) 4 (lane >> 4) + r
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (wave * MW + i) * 16 + 4 * kq + r, n = (j0 + j) * 16 + li;
#ifdef WN_TS_NOATOMIC  // timing experiment only: wrong results
          if (a.rows == 1) a.out[((size_t)m * N + n) * a.ostride + (SPLIT ? tap : b)] = acc[b][i][j][r];
#else
          atomicAdd(a.out + ((size_t)m * N + n) * a.ostride + (SPLIT ? tap : b), acc[b][i][j][r]);
#endif
        }
  if (do_csum) atomicAdd(a.colsum + tid, csum);
}

template <int C, bool DUAL, int NPW>
void launch_ts_wgrad_npw(TsArgs a, hipStream_t s) {
  constexpr int NSPLIT = NPW < C / 16 ? (DUAL ? 2 : 1) * (C / 16) / NPW : 1;
  constexpr int KB = NSPLIT > 1 ? 32 : 16;  // (the kernel's chunk)
  const size_t n_chunks = (a.rows + KB - 1) / KB;
  size_t slices = NSPLIT == 1 ? 512 : (512 / NSPLIT + 7) / 8 * 8;  // ~two workgroups per CU in all
  if (slices > n_chunks) slices = n_chunks;
  a.chunks_per_wg = (n_chunks + slices - 1) / slices;
  slices = (n_chunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  const size_t grid = NSPLIT == 1 ? slices : 8 * NSPLIT * ((slices + 7) / 8);  // (slices past the last chunk return at once)
  if (operand_bf16()) hipLaunchKernelGGL((wn_ts_wgrad_kernel<C, DUAL, true, NPW>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((wn_ts_wgrad_kernel<C, DUAL, false, NPW>), dim3((unsigned)grid), dim3(256), 0, s, a);
}

// column tiles per workgroup: all of them (operands stream exactly once) while every workgroup has >= 64 chunks of 16 rows to
// amortise its [2C, C(, 2)] atomic adds over; half or one of them for fewer rows.  env BLVM_WN_WGRAD_NPW forces a value (experiments)
template <int C, bool DUAL>
int launch_ts_wgrad(TsArgs a, hipStream_t s) {
  constexpr int NT = C / 16;
  const size_t n_chunks = (a.rows + 15) / 16;
  const char* fe = getenv("BLVM_WN_WGRAD_NPW");  // read per call: the parity test runs both forms on the same inputs in one process
  const int forced = fe ? atoi(fe) : 0;
  int npw = n_chunks >= 512 * 64 ? NT : (NT >= 2 ? NT / 2 : 1);
  if (forced > 0) npw = forced >= NT ? NT : (forced > 1 && NT >= 2 ? NT / 2 : 1);
  if (npw >= NT) launch_ts_wgrad_npw<C, DUAL, NT>(a, s);
  else if (npw > 1) launch_ts_wgrad_npw<C, DUAL, (NT >= 2 ? NT / 2 : 1)>(a, s);
  else launch_ts_wgrad_npw<C, DUAL, 1>(a, s);
  return BLVM_OK;
}

struct ConvWs { float *W0, *W1, *dW0, *dW1; };

}  // namespace

// out[L_out*B, N] = x[0:L_out*B] W0^T + x[d*B : d*B + L_out*B] W1^T + bias, with (W0, W1) already split
static int conv_k2_apply(const float* x, int Cin, const float* W0, const float* W1, const float* bias, size_t rows_out,
                         size_t shift_rows, int N, float* out, hipStream_t s) {
  int rc = gemm_f32(0, 0, (int)rows_out, N, Cin, x, Cin, W0, Cin, out, N, bias, 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  return gemm_f32(0, 0, (int)rows_out, N, Cin, x + shift_rows * Cin, Cin, W1, Cin, out, N, nullptr, 0, 0.f, nullptr, 0, 1, 1, s);
}

}  // namespace blvm

using namespace blvm;

extern "C" int blvm_scale_act_f32(const float* x, float scale, float slope, float* y, size_t n, void* stream) {
  if (n == 0) return BLVM_OK;
  BLVM_REQUIRE(x && y, "scale_act: null pointer");
  hipLaunchKernelGGL(scale_act_kernel, ew_grid(n), dim3(256), 0, static_cast<hipStream_t>(stream), x, scale, slope, y, n);
  BLVM_CHECK_LAUNCH("scale_act_f32");
  return BLVM_OK;
}

// ---- plain dilated causal convolution, kernel size 2 ------------------------------------------------------------------
extern "C" size_t blvm_conv1d_k2_workspace_floats(int Cin, int Cout) { return (size_t)4 * Cin * Cout + 16; }

extern "C" int blvm_conv1d_k2_fwd(const float* x, const float* W, const float* bias, int L_in, int B, int Cin, int Cout,
                                  int dilation, float* out, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(x && W && out && workspace, "conv1d_k2_fwd: null pointer");
  BLVM_REQUIRE(L_in > dilation && dilation > 0 && B > 0 && Cin > 0 && Cout > 0, "conv1d_k2_fwd: bad shape");
  const size_t nk = (size_t)Cin * Cout;
  float *W0 = workspace, *W1 = workspace + ((nk + 3) & ~(size_t)3);
  hipLaunchKernelGGL(split_taps_kernel, ew_grid(nk), dim3(256), 0, s, W, W0, W1, nk);
  BLVM_CHECK_LAUNCH("split_taps");
  return conv_k2_apply(x, Cin, W0, W1, bias, (size_t)(L_in - dilation) * B, (size_t)dilation * B, Cout, out, s);
}

extern "C" int blvm_conv1d_k2_bwd(const float* x, const float* W, const float* d_out, int L_in, int B, int Cin, int Cout,
                                  int dilation, float* d_x, float* dW, float* db, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(x && W && d_out && workspace, "conv1d_k2_bwd: null pointer");
  BLVM_REQUIRE(L_in > dilation && dilation > 0 && B > 0 && Cin > 0 && Cout > 0, "conv1d_k2_bwd: bad shape");
  const size_t nk = (size_t)Cin * Cout, nk4 = (nk + 3) & ~(size_t)3;
  float *W0 = workspace, *W1 = W0 + nk4, *dW0 = W1 + nk4, *dW1 = dW0 + nk4;
  const size_t rows = (size_t)(L_in - dilation) * B, shift = (size_t)dilation * B;
  int rc;
  if (d_x) {
    hipLaunchKernelGGL(split_taps_kernel, ew_grid(nk), dim3(256), 0, s, W, W0, W1, nk);
    BLVM_HIP(hipMemsetAsync(d_x, 0, sizeof(float) * (size_t)L_in * B * Cin, s));
    rc = gemm_f32(0, 1, (int)rows, Cin, Cout, d_out, Cout, W0, Cin, d_x, Cin, nullptr, 0, 0.f, nullptr, 0, 1, 1, s);
    if (rc) return rc;
    rc = gemm_f32(0, 1, (int)rows, Cin, Cout, d_out, Cout, W1, Cin, d_x + shift * Cin, Cin, nullptr, 0, 0.f, nullptr, 0, 1, 1, s);
    if (rc) return rc;
  }
  if (dW) {
    BLVM_HIP(hipMemsetAsync(dW0, 0, sizeof(float) * 2 * nk4, s));
    const int sp = pick_split(Cout, Cin, (int)rows);
    rc = gemm_f32(1, 1, Cout, Cin, (int)rows, d_out, Cout, x, Cin, dW0, Cin, nullptr, 0, 0.f, nullptr, 0, 1, sp, s, db);  // (+ db)
    if (rc) return rc;
    rc = gemm_f32(1, 1, Cout, Cin, (int)rows, d_out, Cout, x + shift * Cin, Cin, dW1, Cin, nullptr, 0, 0.f, nullptr, 0, 1, sp, s);
    if (rc) return rc;
    hipLaunchKernelGGL(merge_taps_kernel, ew_grid(nk), dim3(256), 0, s, dW0, dW1, dW, nk);
  }
  if (db && !dW) { rc = colsum_f32((int)rows, Cout, d_out, Cout, db, 1, s); if (rc) return rc; }
  BLVM_CHECK_LAUNCH("conv1d_k2_bwd");
  return BLVM_OK;
}

// ---- gated residual block ------------------------------------------------------------------------------------------------
extern "C" size_t blvm_wavenet_block_reserve_floats(int L_in, int B, int C, int dilation) {
  return (size_t)(L_in - dilation) * B * 3 * C + 16;  // pre [rows,2C] + act [rows,C]
}
extern "C" size_t blvm_wavenet_block_workspace_floats(int L_in, int B, int C, int S, int dilation) {
  const size_t rows = (size_t)(L_in - dilation) * B;
  // W0,W1,dW0,dW1 [2C*C each] + rs/d_rs [rows,C+S] + d_act [rows,C] + d_pre [rows,2C]
  return (size_t)8 * C * C + 64 + rows * (C + S) + rows * C + rows * 2 * C;
}

extern "C" int blvm_wavenet_block_fwd(const float* x, const float* conv_w, const float* conv_b, const float* rs_w,
                                      const float* rs_b, int L_in, int B, int C, int S, int dilation, int T_skip,
                                      float inv_std, float* o, float* skip, float* reserve, float* workspace,
                                      void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(x && conv_w && conv_b && rs_w && rs_b && (skip || S == 0) && reserve && workspace, "wavenet_block_fwd: null pointer");
  BLVM_REQUIRE(C > 0 && C % 4 == 0 && S >= 0 && L_in > dilation && dilation > 0 && B > 0, "wavenet_block_fwd: bad shape");
  const int L_out = L_in - dilation;
  BLVM_REQUIRE(T_skip > 0 && T_skip <= L_out, "wavenet_block_fwd: skip size %d exceeds the block's output length %d", T_skip, L_out);
  BLVM_REQUIRE(aligned16(x) && aligned16(reserve) && aligned16(workspace), "wavenet_block_fwd: buffers must be 16-byte aligned");
  const size_t rows = (size_t)L_out * B, shift = (size_t)dilation * B, nk = (size_t)2 * C * C;
  float *pre = reserve, *act = reserve + rows * 2 * C;
  float *W0 = workspace, *W1 = W0 + nk, *rs = W1 + 3 * nk + 64;
  if (fused_fwd_enabled() && S == C && (C == 32 || C == 64 || C == 96) && aligned16(skip) && (o == nullptr || aligned16(o))) {
    // operand-layout copies: the two taps straight out of the interleaved Conv1d weight, the 1x1 weight behind them
    float* Wrs = W1 + nk;
    T16PackScope pack_scope(false, s);  // the three packs in one launch (the block kernels round fp32 packs themselves in the bf16 mode)
    int rc = t16_pack(conv_w, 2 * C, 2, 2 * C, C, W0, s); if (rc) return rc;
    rc = t16_pack(conv_w + 1, 2 * C, 2, 2 * C, C, W1, s); if (rc) return rc;
    rc = t16_pack_rows(rs_w, C, C + S, C, Wrs, s); if (rc) return rc;
    rc = pack_scope.flush(); if (rc) return rc;
    FusedFwdArgs a;
    a.x = x; a.W0 = W0; a.W1 = W1; a.Wrs = Wrs; a.conv_b = conv_b; a.rs_b = rs_b;
    a.pre = pre; a.act = act; a.o = o; a.skip = skip;
    a.rows = rows; a.shift = shift; a.off = rows - (size_t)T_skip * B; a.inv_std = inv_std;
    rc = C == 32 ? launch_fused_fwd<32, 32>(a, s) : C == 64 ? launch_fused_fwd<64, 64>(a, s) : launch_fused_fwd<96, 96>(a, s);
    if (rc) return rc;
    BLVM_CHECK_LAUNCH("wavenet_block_fwd (fused)");
    return BLVM_OK;
  }
  hipLaunchKernelGGL(split_taps_kernel, ew_grid(nk), dim3(256), 0, s, conv_w, W0, W1, nk);
  int rc = conv_k2_apply(x, C, W0, W1, conv_b, rows, shift, 2 * C, pre, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gate_fwd_kernel, ew_grid(rows * (C / 4)), dim3(256), 0, s, pre, act, rows, C);
  rc = gemm_f32(0, 0, (int)rows, C + S, C, act, C, rs_w, C, rs, C + S, rs_b, 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  hipLaunchKernelGGL(resskip_fwd_kernel, ew_grid(rows * (C + S)), dim3(256), 0, s, rs, x + shift * C, o, skip, rows,
                     rows - (size_t)T_skip * B, C, S, inv_std);
  BLVM_CHECK_LAUNCH("wavenet_block_fwd");
  return BLVM_OK;
}

extern "C" int blvm_wavenet_block_bwd(const float* x, const float* conv_w, const float* rs_w, const float* reserve,
                                      const float* d_o, const float* d_skip, int L_in, int B, int C, int S, int dilation,
                                      int T_skip, float inv_std, float* d_x, float* dconv_w, float* dconv_b, float* drs_w,
                                      float* drs_b, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(x && conv_w && rs_w && reserve && (d_skip || S == 0) && d_x && workspace, "wavenet_block_bwd: null pointer");
  BLVM_REQUIRE(C > 0 && C % 4 == 0 && S >= 0 && L_in > dilation && dilation > 0 && B > 0, "wavenet_block_bwd: bad shape");
  const int L_out = L_in - dilation;
  BLVM_REQUIRE(T_skip > 0 && T_skip <= L_out, "wavenet_block_bwd: bad skip size");
  BLVM_REQUIRE(aligned16(x) && aligned16(reserve) && aligned16(workspace) && aligned16(d_x), "wavenet_block_bwd: alignment");
  const size_t rows = (size_t)L_out * B, shift = (size_t)dilation * B, nk = (size_t)2 * C * C;
  const float *pre = reserve, *act = reserve + rows * 2 * C;
  float *W0 = workspace, *W1 = W0 + nk, *dW0 = W1 + nk, *dW1 = dW0 + nk;
  float* d_rs = dW1 + nk + 64;
  float* d_act = d_rs + rows * (C + S);
  float* d_pre = d_act + rows * C;
  int rc;
  const bool fused = fused_fwd_enabled() && S == C && (C == 32 || C == 64 || C == 96) && (d_o == nullptr || aligned16(d_o)) && aligned16(d_skip);
  if (fused) {
    // operand-layout copies, 2 C^2 floats each: Wrs^T [C, C+S] in the W0 slot, the taps' transposes [C, 2C] in the W1 and
    // dW0 slots (dW0 is zero-filled for the weight-gradient GEMMs further down the stream, after kernel B has read it)
    float* WrsT = W0;
    float* W0T = W1;
    float* W1T = dW0;
    T16PackScope pack_scope(false, s);
    rc = t16_pack_transposed(rs_w, C, C + S, C, WrsT, s); if (rc) return rc;
    rc = t16_pack(conv_w, 2, 2 * C, C, 2 * C, W0T, s); if (rc) return rc;
    rc = t16_pack(conv_w + 1, 2, 2 * C, C, 2 * C, W1T, s); if (rc) return rc;
    rc = pack_scope.flush(); if (rc) return rc;
    FusedBwdAArgs aa;
    aa.d_o = d_o; aa.d_skip = d_skip; aa.pre = pre; aa.WrsT = WrsT; aa.d_rs = d_rs; aa.d_pre = d_pre;
    aa.rows = rows; aa.off = rows - (size_t)T_skip * B; aa.inv_std = inv_std;
    FusedBwdBArgs ab;
    ab.d_pre = d_pre; ab.d_o = d_o; ab.W0T = W0T; ab.W1T = W1T; ab.d_x = d_x; ab.rows = rows; ab.shift = shift; ab.inv_std = inv_std;
    rc = C == 32 ? launch_fused_bwd<32, 32>(aa, ab, s) : C == 64 ? launch_fused_bwd<64, 64>(aa, ab, s) : launch_fused_bwd<96, 96>(aa, ab, s);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(split_taps_kernel, ew_grid(nk), dim3(256), 0, s, conv_w, W0, W1, nk);
    // d_x: rows [0, d*B) start at zero, rows [d*B, L*B) start with the residual path d_o * inv_std
    BLVM_HIP(hipMemsetAsync(d_x, 0, sizeof(float) * shift * C, s));
    hipLaunchKernelGGL(resskip_bwd_kernel, ew_grid(rows * (C + S)), dim3(256), 0, s, d_o, d_skip, d_rs, d_x + shift * C, rows,
                       rows - (size_t)T_skip * B, C, S, inv_std);
    // 1x1 convolution
    rc = gemm_f32(0, 1, (int)rows, C, C + S, d_rs, C + S, rs_w, C, d_act, C, nullptr, 0, 0.f, nullptr, 0, 0, 1, s);
    if (rc) return rc;
  }
  if (fused) {
    // weight and bias gradients: the operands stream once (wn_ts_wgrad_kernel); the taps land interleaved in dconv_w
    if (drs_w || drs_b) {
      TsArgs t{};
      t.A = d_rs; t.B0 = act; t.B1 = nullptr; t.rows = rows; t.out = drs_w; t.ostride = 1; t.colsum = drs_b;
      BLVM_REQUIRE(drs_w != nullptr, "wavenet_block_bwd: drs_b without drs_w");
      rc = C == 32 ? launch_ts_wgrad<32, false>(t, s) : C == 64 ? launch_ts_wgrad<64, false>(t, s) : launch_ts_wgrad<96, false>(t, s);
      if (rc) return rc;
    }
    if (dconv_w || dconv_b) {
      TsArgs t{};
      t.A = d_pre; t.B0 = x; t.B1 = x + shift * C; t.rows = rows; t.out = dconv_w; t.ostride = 2; t.colsum = dconv_b;
      BLVM_REQUIRE(dconv_w != nullptr, "wavenet_block_bwd: dconv_b without dconv_w");
      rc = C == 32 ? launch_ts_wgrad<32, true>(t, s) : C == 64 ? launch_ts_wgrad<64, true>(t, s) : launch_ts_wgrad<96, true>(t, s);
      if (rc) return rc;
    }
    BLVM_CHECK_LAUNCH("wavenet_block_bwd (fused)");
    return BLVM_OK;
  }
  // (bias gradients = column sums of the GEMM's A operand: they ride in its first column block, gemm.hip)
  if (drs_w) { rc = gemm_f32(1, 1, C + S, C, (int)rows, d_rs, C + S, act, C, drs_w, C, nullptr, 0, 0.f, nullptr, 0, 1, pick_split(C + S, C, (int)rows), s, drs_b); if (rc) return rc; }
  else if (drs_b) { rc = colsum_f32((int)rows, C + S, d_rs, C + S, drs_b, 1, s); if (rc) return rc; }
  // gate
  if (!fused) hipLaunchKernelGGL(gate_bwd_kernel, ew_grid(rows * (C / 4)), dim3(256), 0, s, pre, d_act, d_pre, rows, C);
  // dilated convolution: weight gradients of both taps, then the two shifted input gradients
  if (dconv_w) {
    BLVM_HIP(hipMemsetAsync(dW0, 0, sizeof(float) * 2 * nk, s));
    const int sp = pick_split(2 * C, C, (int)rows);
    rc = gemm_f32(1, 1, 2 * C, C, (int)rows, d_pre, 2 * C, x, C, dW0, C, nullptr, 0, 0.f, nullptr, 0, 1, sp, s, dconv_b);
    if (rc) return rc;
    rc = gemm_f32(1, 1, 2 * C, C, (int)rows, d_pre, 2 * C, x + shift * C, C, dW1, C, nullptr, 0, 0.f, nullptr, 0, 1, sp, s);
    if (rc) return rc;
    hipLaunchKernelGGL(merge_taps_kernel, ew_grid(nk), dim3(256), 0, s, dW0, dW1, dconv_w, nk);
  }
  if (dconv_b && !dconv_w) { rc = colsum_f32((int)rows, 2 * C, d_pre, 2 * C, dconv_b, 1, s); if (rc) return rc; }
  if (!fused) {
    rc = gemm_f32(0, 1, (int)rows, C, 2 * C, d_pre, 2 * C, W0, C, d_x, C, nullptr, 0, 0.f, nullptr, 0, 1, 1, s);
    if (rc) return rc;
    rc = gemm_f32(0, 1, (int)rows, C, 2 * C, d_pre, 2 * C, W1, C, d_x + shift * C, C, nullptr, 0, 0.f, nullptr, 0, 1, 1, s);
    if (rc) return rc;
  }
  BLVM_CHECK_LAUNCH("wavenet_block_bwd");
  return BLVM_OK;
}

// ---- the whole residual stack in one call (`ResidualStack.forward`, wavenet_modules.py:178-215) ---------------------------------
// At the reference's own batch (4 utterances) a block's kernels take 0.1-0.2 ms while the Python loop around blvm_wavenet_block_*
// (two allocations and seventeen converted arguments per block) takes as long: the host, not the GPU, set the step time.  Here the
// host loop is in the library; block outputs and reserves are slices of two caller-allocated buffers.
namespace {
// block i (input length Li): output rows and reserve floats, in floats from the start of the concatenated buffers
struct StackLayout {
  std::vector<size_t> act_off, res_off;  // [n + 1]
  std::vector<int> L_in;                 // [n]
};
int stack_layout(int L, int B, int C, const int* dilations, int n, StackLayout& lay) {
  BLVM_REQUIRE(n > 0 && n <= 4096 && dilations != nullptr && L > 0 && B > 0 && C > 0 && C % 4 == 0, "wavenet_stack: bad shape (%d blocks)", n);
  lay.act_off.assign(n + 1, 0); lay.res_off.assign(n + 1, 0); lay.L_in.assign(n, 0);
  int Li = L;
  for (int i = 0; i < n; ++i) {
    BLVM_REQUIRE(dilations[i] > 0 && Li > dilations[i], "wavenet_stack: block %d: input length %d, dilation %d", i, Li, dilations[i]);
    lay.L_in[i] = Li;
    lay.res_off[i + 1] = lay.res_off[i] + blvm_wavenet_block_reserve_floats(Li, B, C, dilations[i]);
    lay.act_off[i + 1] = lay.act_off[i] + (i + 1 < n ? (size_t)(Li - dilations[i]) * B * C : 0);  // the last block has no residual output
    Li -= dilations[i];
  }
  return BLVM_OK;
}
}  // namespace

extern "C" int blvm_wavenet_stack_floats(int L, int B, int C, const int* dilations, int n_blocks, size_t* acts_floats, size_t* reserve_floats) {
  StackLayout lay;
  const int rc = stack_layout(L, B, C, dilations, n_blocks, lay);
  if (rc) return rc;
  if (acts_floats) *acts_floats = lay.act_off[n_blocks];
  if (reserve_floats) *reserve_floats = lay.res_off[n_blocks];
  return BLVM_OK;
}

extern "C" int blvm_wavenet_stack_fwd(const float* x, const float* const* params, const int* dilations, const int* groups, int n_blocks, int L,
                                      int B, int C, int S, int T_skip, float inv_std, float* acts, float* const* skips, float* reserve,
                                      float* workspace, void* stream) {
  BLVM_REQUIRE(x && params && groups && reserve && workspace && (acts || n_blocks == 1), "wavenet_stack_fwd: null pointer");
  StackLayout lay;
  int rc = stack_layout(L, B, C, dilations, n_blocks, lay);
  if (rc) return rc;
  for (int i = 0; i < n_blocks; ++i) {
    const float* xi = i == 0 ? x : acts + lay.act_off[i - 1];
    float* o = i + 1 < n_blocks ? acts + lay.act_off[i] : nullptr;
    const int Si = groups[i] >= 0 ? S : 0;
    BLVM_REQUIRE(Si == 0 || (skips && skips[groups[i]]), "wavenet_stack_fwd: block %d: no skip tensor %d", i, groups[i]);
    rc = blvm_wavenet_block_fwd(xi, params[4 * i], params[4 * i + 1], params[4 * i + 2], params[4 * i + 3], lay.L_in[i], B, C, Si, dilations[i],
                                T_skip, inv_std, o, Si ? skips[groups[i]] : nullptr, reserve + lay.res_off[i], workspace, stream);
    if (rc) return rc;
  }
  return BLVM_OK;
}

extern "C" int blvm_wavenet_stack_bwd(const float* x, const float* const* params, const int* dilations, const int* groups, int n_blocks, int L,
                                      int B, int C, int S, int T_skip, float inv_std, const float* acts, const float* reserve,
                                      const float* const* d_skips, float* d_x, float* d_scratch, float* const* grads, float* workspace,
                                      void* stream) {
  BLVM_REQUIRE(x && params && groups && reserve && workspace && d_x && grads && (acts || n_blocks == 1) && (d_scratch || n_blocks == 1),
               "wavenet_stack_bwd: null pointer");
  StackLayout lay;
  int rc = stack_layout(L, B, C, dilations, n_blocks, lay);
  if (rc) return rc;
  // input gradients ping-pong between d_x (even blocks; block 0's is the result) and d_scratch (odd blocks), both [L,B,C]
  const float* d_o = nullptr;
  for (int i = n_blocks - 1; i >= 0; --i) {
    const float* xi = i == 0 ? x : acts + lay.act_off[i - 1];
    float* d_xi = (i & 1) ? d_scratch : d_x;
    const int Si = groups[i] >= 0 ? S : 0;
    BLVM_REQUIRE(Si == 0 || (d_skips && d_skips[groups[i]]), "wavenet_stack_bwd: block %d: no skip gradient %d", i, groups[i]);
    rc = blvm_wavenet_block_bwd(xi, params[4 * i], params[4 * i + 2], reserve + lay.res_off[i], d_o, Si ? d_skips[groups[i]] : nullptr, lay.L_in[i],
                                B, C, Si, dilations[i], T_skip, inv_std, d_xi, grads[4 * i], grads[4 * i + 1], grads[4 * i + 2], grads[4 * i + 3],
                                workspace, stream);
    if (rc) return rc;
    d_o = d_xi;
  }
  return BLVM_OK;
}
