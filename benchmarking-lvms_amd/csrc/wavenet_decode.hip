// wavenet_decode.hip — K10c: autoregressive WaveNet sampling, every frame of every utterance in ONE launch.
//
// Replaces the per-frame loop of WaveNet.generate (blvm/models/wavenet/wavenet.py:254-293).  The reference re-runs the
// causal convolution and all residual blocks over a receptive-field window for each new sample; its TODO names the cached
// formulation (arXiv:1611.09482), which this kernel is: block i keeps a ring buffer of its own input over the last
// `dilation_i` frames, so one new frame costs one [2C]x[2C] and one [C+S]x[C] product per block.
//
// Decoding is a chain of ~2 * n_blocks dependent small products per frame — pure latency.  One workgroup owns 16
// utterances (the M dimension of v_mfma_f32_16x16x4_f32) and walks the whole network for them with all activations in
// LDS; the only global traffic is the weights (which stay L2 / Infinity-Cache resident across frames, the kernel is never
// left), one ring-buffer row per block, the two uniform draws and the sample.  No launch, no host, no inter-workgroup
// synchronisation inside the loop.
//
// A window of zeros is an all-zero past; under it every block input is constant in time (the network's response to zero
// input, biases included).  The prologue evaluates that steady state once — each block with both taps on the same vector —
// and fills the ring buffers with it, which makes frame 0 identical to the reference's first window evaluation.
#include "common.h"

namespace blvm {
namespace {

constexpr int DEC_ROWS = 16;        // utterances per workgroup
constexpr int DEC_MAX_BLOCKS = 64;  // dilations travel by value in the kernel arguments
constexpr int DEC_HEAD_ROWS = 32;   // head weight rows in the packed image (3 * num_mix <= 30, zero padded)

// Offsets (floats) into the packed weight image; every region starts on a 16-byte boundary (C, S, O multiples of 16).
struct DecodeLayout {
  size_t causal_w, causal_b, in_w, in_b, blocks, block_stride, out_w, out_b, head_w, head_b, total;
  size_t conv_b, rs_w, rs_b;  // relative to a block's start (conv_w is at 0)
};

__host__ __device__ inline DecodeLayout decode_layout(int C, int S, int O, int n_blocks) {
  DecodeLayout L;
  size_t o = 0;
  L.causal_w = o; o += (size_t)C * 2;  // [C,1,2]
  L.causal_b = o; o += C;
  L.in_w = o; o += (size_t)C * C;
  L.in_b = o; o += C;
  L.conv_b = (size_t)2 * C * 2 * C;
  L.rs_w = L.conv_b + 2 * C;
  L.rs_b = L.rs_w + (size_t)(C + S) * C;
  L.block_stride = L.rs_b + (C + S);
  L.blocks = o; o += L.block_stride * n_blocks;
  L.out_w = o; o += (size_t)O * S;
  L.out_b = o; o += O;
  L.head_w = o; o += (size_t)DEC_HEAD_ROWS * O;
  L.head_b = o; o += DEC_HEAD_ROWS;
  L.total = o;
  return L;
}

struct DecodeArgs {
  const float* w;
  const float* wt;  // T16 operand copies (common.h) of the blocks' matrices: block i at i * ((2C)^2 + (C+S) C): conv [2C,2C], rs [C+S,C]
  int dil[DEC_MAX_BLOCKS];
  int n_blocks, B, C, S, O, n_frames, num_mix;
  float inv_std, skip_scale, log_eps;
  const float* u;  // [n_frames,B,num_mix] or NULL
  const float* v;  // [n_frames,B] or NULL
  float* queues;   // block i: [dil_i,B,C] at B*C*sum(dil[:i])
  float* x_out;    // [B,n_frames]
};

// LDS-only barrier: waits for this wave's LDS traffic, not for its global loads.  No thread of the decoder ever reads
// global memory another thread wrote (ring-buffer elements are read and rewritten by the same thread, weights and draws are
// read-only), so weight / ring-buffer loads issued for the NEXT block stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16x16 output tile from an LDS operand and weights already in registers: wreg[j] holds W[col][16 j + 4 q .. +3]
template <int KCH>
__device__ __forceinline__ f32x4 tile_from_regs(const float* __restrict__ A, int lda, const float4 (&wreg)[KCH]) {
  const int lane = threadIdx.x & 63;
  const float* ap = A + (lane & 15) * lda + 4 * (lane >> 4);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KCH; ++j) {
    const float4 x = *reinterpret_cast<const float4*>(ap + 16 * j);
    f32x4& acc = (j & 1) ? acc1 : acc0;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wreg[j].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wreg[j].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wreg[j].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wreg[j].w, acc, 0, 0, 0);
  }
  return acc0 + acc1;
}

// CC, SS > 0: widths known at compile time — the main loop keeps each wave's weight tiles, biases and ring-buffer taps of
// the NEXT block in registers, loaded while the current block computes.  CC == 0: any width, loads where they are used.
template <int NW, int CC, int SS>
__global__ __launch_bounds__(NW * 64) void wn_decode_kernel(DecodeArgs a) {
  extern __shared__ __align__(16) float smem[];
  constexpr int NT = NW * 64;
  const int C = CC > 0 ? CC : a.C, S = SS > 0 ? SS : a.S, O = a.O, B = a.B;
  const int CA = C > O ? C : O;
  const int ldV = 2 * C + 4, ldA = CA + 4, ldS = S + 4, ldP = 2 * C;
  float* sV = smem;                    // [16][2C+4]  interleaved taps: k = 2c + tap
  float* sPre = sV + DEC_ROWS * ldV;   // [16][2C]    gate pre-activations
  float* sAct = sPre + DEC_ROWS * ldP; // [16][max(C,O)+4]
  float* sH = sAct + DEC_ROWS * ldA;   // [16][C]     current block input
  float* sSkip = sH + DEC_ROWS * C;    // [16][S+4]
  float* sPar = sSkip + DEC_ROWS * ldS;  // [16][32]  head outputs
  float* sX = sPar + DEC_ROWS * DEC_HEAD_ROWS;  // [16][2] previous / newest sample

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, cc = lane & 15;
  const int b0 = blockIdx.x * DEC_ROWS;
  const DecodeLayout L = decode_layout(C, S, O, a.n_blocks);
  const float* w = a.w;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < DEC_ROWS * 2; i += NT) sX[i] = 0.f;
  __syncthreads();

  // causal conv on (previous, newest) sample -> 1x1 in_transform -> sH; clears the skip accumulators
  auto front = [&]() {
    const float* cw = w + L.causal_w;
    const float* cb = w + L.causal_b;
    for (int idx = tid; idx < DEC_ROWS * C; idx += NT) {
      const int r = idx / C, o = idx - r * C;
      sAct[r * ldA + o] = cw[2 * o] * sX[2 * r] + cw[2 * o + 1] * sX[2 * r + 1] + cb[o];
    }
    for (int idx = tid; idx < DEC_ROWS * S; idx += NT) {
      const int r = idx / S;
      sSkip[r * ldS + (idx - r * S)] = 0.f;
    }
    __syncthreads();
    for (int tile = wave; tile < C / 16; tile += NW) {
      const f32x4 acc = wave_gemm16<1>(sAct, ldA, 0, DEC_ROWS, w + L.in_w, C, tile * 16, C, 0, zero4);
      const int o = tile * 16 + cc;
      const float bias = w[L.in_b + o];
#pragma unroll
      for (int r = 0; r < 4; ++r) sH[(4 * q + r) * C + o] = acc[r] + bias;
    }
    __syncthreads();
  };

  // one gated residual block on the frame in sH.  steady: both taps = sH, ring buffer filled with sH, skip untouched.
  const size_t wt_stride = (size_t)2 * C * 2 * C + (size_t)(C + S) * C;
  auto block = [&](int i, float* qi, int slot, bool steady) {
    const float* bw = w + L.blocks + (size_t)i * L.block_stride;
    const float* bt = a.wt + (size_t)i * wt_stride;
    const int d = a.dil[i];
    for (int idx = tid; idx < DEC_ROWS * C; idx += NT) {
      const int r = idx / C, c = idx - r * C;
      const float cur = sH[idx];
      float old = cur;
      if (b0 + r < B) {
        if (steady) {
          for (int s = 0; s < d; ++s) qi[((size_t)s * B + b0 + r) * C + c] = cur;
        } else {
          float* p = qi + ((size_t)slot * B + b0 + r) * C + c;
          old = *p;
          *p = cur;
        }
      }
      sV[r * ldV + 2 * c] = old;
      sV[r * ldV + 2 * c + 1] = cur;
    }
    __syncthreads();
    for (int tile = wave; tile < 2 * C / 16; tile += NW) {
      const f32x4 acc = wave_gemm16<1, true>(sV, ldV, 0, DEC_ROWS, bt, 2 * C, tile * 16, 2 * C, 0, zero4);
      const int o = tile * 16 + cc;
      const float bias = bw[L.conv_b + o];
#pragma unroll
      for (int r = 0; r < 4; ++r) sPre[(4 * q + r) * ldP + o] = acc[r] + bias;
    }
    __syncthreads();
    for (int idx = tid; idx < DEC_ROWS * C; idx += NT) {
      const int r = idx / C, c = idx - r * C;
      sAct[r * ldA + c] = tanhf(sPre[r * ldP + c]) * sigmoidf_(sPre[r * ldP + C + c]);
    }
    __syncthreads();
    for (int tile = wave; tile < (C + S) / 16; tile += NW) {
      const f32x4 acc = wave_gemm16<1, true>(sAct, ldA, 0, DEC_ROWS, bt + (size_t)2 * C * 2 * C, C, tile * 16, C, 0, zero4);
      const int o = tile * 16 + cc;
      const float bias = bw[L.rs_b + o];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        const float val = acc[r] + bias;
        if (o < C) sH[row * C + o] = (val + sH[row * C + o]) * a.inv_std;
        else if (!steady) sSkip[row * ldS + (o - C)] += val;
      }
    }
    __syncthreads();
  };

  // steady state under an all-zero past
  front();
  {
    float* qi = a.queues;
    for (int i = 0; i < a.n_blocks; ++i) {
      block(i, qi, 0, true);
      qi += (size_t)a.dil[i] * B * C;
    }
  }

  // ---- register-resident prefetch state of the compile-time-width path
  constexpr int C_ = CC > 0 ? CC : 16, S_ = SS > 0 ? SS : 16;
  constexpr int NT1 = 2 * C_ / 16, T1 = (NT1 + NW - 1) / NW, K1 = 2 * C_ / 16;  // conv: tiles, tiles per wave, k chunks
  constexpr int NT2 = (C_ + S_) / 16, T2 = (NT2 + NW - 1) / NW, K2 = C_ / 16;   // rs
  constexpr int NQ = (DEC_ROWS * C_ + NT - 1) / NT;                             // ring-buffer elements per thread
  struct BlockRegs {
    float4 wc[T1][K1], wr[T2][K2];
    float bc[T1], br[T2], oldv[NQ];
  };
  BlockRegs R0, R1;  // ping-pong: one block computes from one set while the next block's operands land in the other
  auto load_block = [&](BlockRegs& R, int i, const float* qi, int slot) {
    const float* bw = w + L.blocks + (size_t)i * L.block_stride;
    const float* bt = a.wt + (size_t)i * wt_stride;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int idx = tid + n * NT;
      const int r = idx / C_, c = idx - r * C_;
      R.oldv[n] = (idx < DEC_ROWS * C_ && b0 + r < B) ? qi[((size_t)slot * B + b0 + r) * C_ + c] : 0.f;
    }
#pragma unroll
    for (int tt = 0; tt < T1; ++tt) {
      const int tile = wave + tt * NW;
      if (tile < NT1) {
        const float* wp = bt + (size_t)(tile * 16) * (2 * C_) + 4 * lane;  // T16: chunk j of the tile is one contiguous 1 KB
#pragma unroll
        for (int j = 0; j < K1; ++j) R.wc[tt][j] = *reinterpret_cast<const float4*>(wp + 256 * j);
        R.bc[tt] = bw[L.conv_b + tile * 16 + cc];
      }
    }
#pragma unroll
    for (int tt = 0; tt < T2; ++tt) {
      const int tile = wave + tt * NW;
      if (tile < NT2) {
        const float* wp = bt + (size_t)2 * C_ * 2 * C_ + (size_t)(tile * 16) * C_ + 4 * lane;
#pragma unroll
        for (int j = 0; j < K2; ++j) R.wr[tt][j] = *reinterpret_cast<const float4*>(wp + 256 * j);
        R.br[tt] = bw[L.rs_b + tile * 16 + cc];
      }
    }
  };
#ifdef DEC_PROF
  long long prof[6] = {0, 0, 0, 0, 0, 0};
#define DEC_TICK(k) { const long long now__ = __builtin_readcyclecounter(); prof[k] += now__ - tick__; tick__ = now__; }
#else
#define DEC_TICK(k)
#endif
  // the block of the main loop: operands in `cur` (loaded one block ago); issues every load of the next block
  // (ni, nqi, nslot) into `nxt` right after staging — a whole block ahead of their first use
  auto block_fast = [&](BlockRegs& cur, BlockRegs& nxt, float* qi, int slot, int ni, const float* nqi, int nslot) {
#ifdef DEC_PROF
    long long tick__ = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int idx = tid + n * NT;
      if (idx < DEC_ROWS * C_) {
        const int r = idx / C_, c = idx - r * C_;
        *reinterpret_cast<float2*>(sV + r * ldV + 2 * c) = make_float2(cur.oldv[n], sH[idx]);
      }
    }
    // vmcnt counts loads and stores in order: the ring-buffer store goes out AFTER the last use of a loaded tap, or the
    // wait for the tap would also wait for the store's acknowledgement
    asm volatile("" ::: "memory");
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int idx = tid + n * NT;
      const int r = idx / C_, c = idx - r * C_;
      if (idx < DEC_ROWS * C_ && b0 + r < B) qi[((size_t)slot * B + b0 + r) * C_ + c] = sH[idx];
    }
    load_block(nxt, ni, nqi, nslot);
    DEC_TICK(0)
    lds_barrier();
    DEC_TICK(1)
#pragma unroll
    for (int tt = 0; tt < T1; ++tt) {
      const int tile = wave + tt * NW;
      if (tile < NT1) {
        const f32x4 acc = tile_from_regs<K1>(sV, ldV, cur.wc[tt]);
        const int o = tile * 16 + cc;
#pragma unroll
        for (int r = 0; r < 4; ++r) sPre[(4 * q + r) * ldP + o] = acc[r] + cur.bc[tt];
      }
    }
    DEC_TICK(2)
    lds_barrier();
    DEC_TICK(1)
    for (int idx = tid; idx < DEC_ROWS * C_; idx += NT) {
      const int r = idx / C_, c = idx - r * C_;
      sAct[r * ldA + c] = tanhf(sPre[r * ldP + c]) * sigmoidf_(sPre[r * ldP + C_ + c]);
    }
    DEC_TICK(3)
    lds_barrier();
    DEC_TICK(1)
#pragma unroll
    for (int tt = 0; tt < T2; ++tt) {
      const int tile = wave + tt * NW;
      if (tile < NT2) {
        const f32x4 acc = tile_from_regs<K2>(sAct, ldA, cur.wr[tt]);
        const int o = tile * 16 + cc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * q + r;
          const float val = acc[r] + cur.br[tt];
          if (o < C_) sH[row * C_ + o] = (val + sH[row * C_ + o]) * a.inv_std;
          else sSkip[row * ldS + (o - C_)] += val;
        }
      }
    }
    DEC_TICK(4)
    lds_barrier();
    DEC_TICK(1)
  };
  if constexpr (CC > 0) load_block(R0, 0, a.queues, 0);

  for (int t = 0; t < a.n_frames; ++t) {
    front();
    float* qi = a.queues;
    if constexpr (CC > 0) {
      // two blocks per trip (n_blocks is even on this path): the register sets swap roles by position in the code, not by
      // a run-time choice the compiler would turn into register copies behind a full vmcnt wait
      for (int i = 0; i < a.n_blocks; i += 2) {
        const int d0 = a.dil[i], d1 = a.dil[i + 1];
        float* q1 = qi + (size_t)d0 * B * C;
        float* q2 = q1 + (size_t)d1 * B * C;
        const bool last = i + 2 == a.n_blocks;
        const int ni = last ? 0 : i + 2;
        block_fast(R0, R1, qi, t % d0, i + 1, q1, t % d1);
        block_fast(R1, R0, q1, t % d1, ni, last ? a.queues : q2, (last ? t + 1 : t) % a.dil[ni]);
        qi = q2;
      }
    } else {
      for (int i = 0; i < a.n_blocks; ++i) {
        block(i, qi, t % a.dil[i], false);
        qi += (size_t)a.dil[i] * B * C;
      }
    }
    // relu(skip * scale) -> Linear -> relu -> head Linear -> sample
    for (int idx = tid; idx < DEC_ROWS * S; idx += NT) {
      const int r = idx / S, c = idx - r * S;
      sSkip[r * ldS + c] = fmaxf(sSkip[r * ldS + c] * a.skip_scale, 0.f);
    }
    __syncthreads();
    for (int tile = wave; tile < O / 16; tile += NW) {
      const f32x4 acc = wave_gemm16<1>(sSkip, ldS, 0, DEC_ROWS, w + L.out_w, S, tile * 16, S, 0, zero4);
      const int o = tile * 16 + cc;
      const float bias = w[L.out_b + o];
#pragma unroll
      for (int r = 0; r < 4; ++r) sAct[(4 * q + r) * ldA + o] = fmaxf(acc[r] + bias, 0.f);
    }
    __syncthreads();
    for (int tile = wave; tile < DEC_HEAD_ROWS / 16; tile += NW) {
      const f32x4 acc = wave_gemm16<1>(sAct, ldA, 0, DEC_ROWS, w + L.head_w, O, tile * 16, O, 0, zero4);
      const int o = tile * 16 + cc;
      const float bias = w[L.head_b + o];
#pragma unroll
      for (int r = 0; r < 4; ++r) sPar[(4 * q + r) * DEC_HEAD_ROWS + o] = acc[r] + bias;
    }
    __syncthreads();
    if (tid < DEC_ROWS && b0 + tid < B) {
      // Gumbel-max component pick + clamped logistic draw, as mix_sample_kernel (dmol.hip; blvm/utils/variational.py:309-349)
      const float* p = sPar + tid * DEC_HEAD_ROWS;
      const int K = a.num_mix;
      const size_t f = (size_t)t * B + b0 + tid;
      int best = 0;
      float bv = -INFINITY;
      for (int m = 0; m < K; ++m) {
        float s = p[m];
        if (a.u != nullptr) s -= logf(-logf(a.u[f * K + m]));
        if (s > bv) { bv = s; best = m; }
      }
      const float loc = p[K + best], raw = p[2 * K + best];
      float x = loc;
      if (a.v != nullptr) {
        const float vv = a.v[f];
        x = loc + expf(fmaxf(raw, a.log_eps)) * (logf(vv) - logf(1.f - vv));
        x = fminf(fmaxf(x, -1.f), 1.f);
      }
      a.x_out[(size_t)(b0 + tid) * a.n_frames + t] = x;
      sX[2 * tid] = sX[2 * tid + 1];
      sX[2 * tid + 1] = x;
    }
    __syncthreads();
  }
#ifdef DEC_PROF
  if (blockIdx.x == 0 && (tid == 0 || tid == NT - 1))
    for (int k = 0; k < 6; ++k) a.x_out[(size_t)B * a.n_frames + (tid ? 6 : 0) + k] = (float)(prof[k] / 1000);
#endif
}

inline size_t decode_lds_bytes(int C, int S, int O) {
  const int CA = C > O ? C : O;
  return sizeof(float) * ((size_t)DEC_ROWS * ((2 * C + 4) + 2 * C + (CA + 4) + C + (S + 4) + DEC_HEAD_ROWS) + 2 * DEC_ROWS);
}

}  // namespace
}  // namespace blvm

extern "C" size_t blvm_wavenet_decode_pack_floats(int C, int S, int O, int n_blocks) {
  if (C <= 0 || S <= 0 || O <= 0 || n_blocks <= 0) return 0;
  return blvm::decode_layout(C, S, O, n_blocks).total;
}

static size_t decode_ring_floats(const int* dilations, int n_blocks, int B, int C) {
  size_t n = 0;
  for (int i = 0; i < n_blocks; ++i) n += (size_t)(dilations[i] > 0 ? dilations[i] : 0);
  return n * B * C;
}

extern "C" size_t blvm_wavenet_decode_scratch_floats(const int* dilations, int n_blocks, int B, int C, int S) {
  if (!dilations || n_blocks <= 0 || B <= 0 || C <= 0 || S <= 0) return 0;
  return decode_ring_floats(dilations, n_blocks, B, C) + (size_t)n_blocks * ((size_t)2 * C * 2 * C + (size_t)(C + S) * C);
}

extern "C" int blvm_wavenet_decode(const float* packed, const int* dilations, int n_blocks, int B, int C, int S, int O,
                                   int num_mix, int n_frames, float inv_std, float skip_scale, float log_eps,
                                   const float* u, const float* v, float* scratch, float* x_out, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(packed && dilations && scratch && x_out && aligned16(packed) && aligned16(scratch), "wavenet_decode: NULL or misaligned argument");
  BLVM_REQUIRE(B > 0 && n_frames >= 0 && n_blocks > 0 && n_blocks <= DEC_MAX_BLOCKS, "wavenet_decode: need B > 0, 1 <= n_blocks <= %d", DEC_MAX_BLOCKS);
  BLVM_REQUIRE(C > 0 && S > 0 && O > 0 && C % 16 == 0 && S % 16 == 0 && O % 16 == 0, "wavenet_decode: C, S, O must be multiples of 16");
  BLVM_REQUIRE(num_mix > 0 && 3 * num_mix <= DEC_HEAD_ROWS, "wavenet_decode: num_mix must be in [1, %d]", DEC_HEAD_ROWS / 3);
  BLVM_REQUIRE((u == nullptr) == (v == nullptr), "wavenet_decode: u and v are given together (both NULL: the mode)");
  const size_t lds = decode_lds_bytes(C, S, O);
  BLVM_REQUIRE(lds <= 160 * 1024, "wavenet_decode: C=%d, S=%d, O=%d need %zu bytes of LDS (> 160 KB)", C, S, O, lds);
  if (n_frames == 0) return BLVM_OK;
  DecodeArgs a;
  a.w = packed;
  for (int i = 0; i < n_blocks; ++i) {
    BLVM_REQUIRE(dilations[i] >= 1, "wavenet_decode: dilation %d of block %d", dilations[i], i);
    a.dil[i] = dilations[i];
  }
  for (int i = n_blocks; i < DEC_MAX_BLOCKS; ++i) a.dil[i] = 1;
  a.n_blocks = n_blocks; a.B = B; a.C = C; a.S = S; a.O = O; a.n_frames = n_frames; a.num_mix = num_mix;
  a.inv_std = inv_std; a.skip_scale = skip_scale; a.log_eps = log_eps;
  a.u = u; a.v = v; a.x_out = x_out;
  // scratch = [T16 operand copies of the blocks' matrices | ring buffers]
  const DecodeLayout L = decode_layout(C, S, O, n_blocks);
  const size_t wt_stride = (size_t)2 * C * 2 * C + (size_t)(C + S) * C;
  for (int i = 0; i < n_blocks; ++i) {
    const float* bw = packed + L.blocks + (size_t)i * L.block_stride;
    float* bt = scratch + (size_t)i * wt_stride;
    int rc = t16_pack_rows(bw, 2 * C, 2 * C, 2 * C, bt, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    rc = t16_pack_rows(bw + L.rs_w, C, C + S, C, bt + (size_t)2 * C * 2 * C, static_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  a.wt = scratch;
  a.queues = scratch + (size_t)n_blocks * wt_stride;
  constexpr int NW = 8;
  auto kern = wn_decode_kernel<NW, 0, 0>;
  if (n_blocks % 2 == 0 && C == 64 && S == 64) kern = wn_decode_kernel<NW, 64, 64>;
  else if (n_blocks % 2 == 0 && C == 32 && S == 32) kern = wn_decode_kernel<NW, 32, 32>;
  if (lds > 64 * 1024) BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)((B + DEC_ROWS - 1) / DEC_ROWS)), dim3(NW * 64), lds, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("wavenet_decode");
  return BLVM_OK;
}
