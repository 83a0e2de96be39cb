// stages.h — stage kernels shared by the recurrent-cell chains (vrnn.hip, srnn.hip): the generic multi-segment linear
// link, the Gaussian-heads link and the dz/KL link, plus their host-side launch helpers.  Included inside
// `namespace blvm { namespace {` of each translation unit.
#pragma once
#include <vector>

// ---------------------------------------------------------------------------------------------------------------
// Launch-latency notes (measured, profiles/r01_*): a link of the chain costs the kernel boundary (~1.5 us) plus every
// DEPENDENT memory round trip inside the kernel.  So: (1) no dynamically indexed kernel arguments (each would be a
// dependent scalar load) — segments are selected with scalar selects on values that arrive with the first kernarg
// fetch; (2) every epilogue operand (bias, addend, gate, saved activations) is loaded BEFORE the K loop so it
// travels with the operand loads; (3) the K split is as wide as the reduction allows (NW = 4/8/16 waves) so a
// wave's dependent MFMA chain is <= 24 instructions.
// ---------------------------------------------------------------------------------------------------------------

// EVERY weight operand of the stage kernels (W, Wp, Wq, WT ...) is in the T16 operand layout (common.h: t16_pack /
// wave_gemm16<NW, true>), packed once per sequence by the host code of the chain; `ldw` is the packed matrix's K.
//
// generic multi-segment linear stage:  out = gate( act( A W^T + bias + add ) )
// Struct-of-arrays + scalar selects: every field is a plain kernarg scalar (s_load -> SGPR -> s_cselect); absent
// operands are replaced on the host by a valid dummy pointer (W) plus a flag bit, so all prefetches are unconditional.
enum { LF_BIAS = 1, LF_ADD = 2, LF_GATE = 4, LF_RELU = 8 };

template <int NSEG>
struct LinArgs {
  const float* A[NSEG];     // [B,K]
  const float* W[NSEG];     // [ncols,K] in T16
  const float* bias[NSEG];  // [ncols]
  const float* add[NSEG];   // [B,ncols] (may alias out)
  const float* gate[NSEG];  // [B,ncols]: result *= (gate > 0)
  float* out[NSEG];         // [B,ncols]
  int lda[NSEG], ldw[NSEG], ldadd[NSEG], ldgate[NSEG], ldo[NSEG], tiles[NSEG], flags[NSEG], K[NSEG];
  int B;
  float slope;  // negative-side slope of the activation AND of the gate (0 = ReLU, 0.01 = LeakyReLU)
};

#define PICK(f) (NSEG == 1 ? a.f[0] : (s == 0 ? a.f[0] : (NSEG == 2 || s == 1 ? a.f[NSEG > 1 ? 1 : 0] : a.f[NSEG > 2 ? 2 : 0])))

template <int NW, int NSEG>
__global__ __launch_bounds__(NW * 64) void lin_stage_kernel(LinArgs<NSEG> a) {
  __shared__ float red[NW * 256];
  int ct = blockIdx.x, s = 0;
  if (NSEG > 1) {
    const int t0 = a.tiles[0], t1 = a.tiles[NSEG > 1 ? 1 : 0];
    s = (ct >= t0 ? 1 : 0) + ((NSEG > 2 && ct >= t0 + t1) ? 1 : 0);
    ct -= (s >= 1 ? t0 : 0) + (s >= 2 ? t1 : 0);
  }
  const float* A = PICK(A);
  const float* W = PICK(W);
  const float* bias = PICK(bias);
  const float* add = PICK(add);
  const float* gate = PICK(gate);
  float* out = PICK(out);
  const int lda = PICK(lda), ldw = PICK(ldw), ldadd = PICK(ldadd), ldgate = PICK(ldgate), ldo = PICK(ldo);
  const int flags = PICK(flags);
  const int K = PICK(K);
  const int r0 = blockIdx.y * 16, c0 = ct * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < a.B;
  const int rowc = row < a.B ? row : r0;  // clamped row: the prefetches below are unconditional
  // epilogue operands first: they travel with the operand loads instead of after the reduction
  const float e_bias = bias[(flags & LF_BIAS) ? col : 0];
  const float e_add = add[(flags & LF_ADD) ? (size_t)rowc * ldadd + col : 0];
  const float e_gate = gate[(flags & LF_GATE) ? (size_t)rowc * ldgate + col : 0];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW, true>(A, lda, r0, a.B, W, ldw, c0, K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (!own) return;
  float x = v[0] + ((flags & LF_BIAS) ? e_bias : 0.f) + ((flags & LF_ADD) ? e_add : 0.f);
  if (flags & LF_RELU) x = x > 0.f ? x : x * a.slope;
  if (flags & LF_GATE) x = e_gate > 0.f ? x : x * a.slope;
  out[(size_t)row * ldo + col] = x;
}

// Multi-segment links with the OPERAND side of their arguments as preloaded scalars (see lin1_stage_kernel below) and the rest —
// epilogue pointers, strides, flags, outputs — in the trailing struct, fetched by s_load while the operand loads are in flight:
// the epilogue prefetch runs in wave_gemm16's `mid` hook.  Two segments: any shapes (16-bit fields).  Three segments: one shared
// lda and K (= ldw), which is what VRNN's first link (three products of h_{t-1}) has; the host falls back to lin_stage_kernel.
template <int NW, int NSEG>
__device__ __forceinline__ void linp_body(const float* A, const float* W, int lda, int ldw, int K, int B, int ct, int s,
                                          const LinArgs<NSEG>& a, float* red) {
  const int r0 = blockIdx.y * 16, c0 = ct * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  float e_bias = 0.f, e_add = 0.f, e_gate = 0.f;
  int flags = 0;
  auto prefetch = [&]() {
    const float* bias = PICK(bias);
    const float* add = PICK(add);
    const float* gate = PICK(gate);
    const int ldadd = PICK(ldadd), ldgate = PICK(ldgate);
    flags = PICK(flags);
    e_bias = bias[(flags & LF_BIAS) ? col : 0];
    e_add = add[(flags & LF_ADD) ? (size_t)rowc * ldadd + col : 0];
    e_gate = gate[(flags & LF_GATE) ? (size_t)rowc * ldgate + col : 0];
  };
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW, true>(A, lda, r0, B, W, ldw, c0, K, threadIdx.x >> 6, acc[0], prefetch);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (!own) return;
  float* out = PICK(out);
  const int ldo = PICK(ldo);
  float x = v[0] + ((flags & LF_BIAS) ? e_bias : 0.f) + ((flags & LF_ADD) ? e_add : 0.f);
  if (flags & LF_RELU) x = x > 0.f ? x : x * a.slope;
  if (flags & LF_GATE) x = e_gate > 0.f ? x : x * a.slope;
  out[(size_t)row * ldo + col] = x;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void linp2_stage_kernel(const float* A0, const float* A1, const float* W0, const float* W1,
                                                              unsigned lda01, unsigned ldw01, unsigned k01, unsigned tiles0_b,
                                                              LinArgs<2> a) {
  __shared__ float red[NW * 256];
  const int tiles0 = tiles0_b & 0xffff, B = tiles0_b >> 16;
  const bool s1 = (int)blockIdx.x >= tiles0;  // uniform
  const int ct = (int)blockIdx.x - (s1 ? tiles0 : 0);
  linp_body<NW, 2>(s1 ? A1 : A0, s1 ? W1 : W0, s1 ? lda01 >> 16 : lda01 & 0xffff, s1 ? ldw01 >> 16 : ldw01 & 0xffff,
                   s1 ? k01 >> 16 : k01 & 0xffff, B, ct, s1 ? 1 : 0, a, red);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void linp3_stage_kernel(const float* A0, const float* A1, const float* A2, const float* W0,
                                                              const float* W1, const float* W2, unsigned lda_k,
                                                              unsigned b_t0_t1, LinArgs<3> a) {
  __shared__ float red[NW * 256];
  const int lda = (lda_k & 0xffff), K = lda_k >> 16;
  const int B = b_t0_t1 & 0xfff, t0 = (b_t0_t1 >> 12) & 0x3ff, t1 = b_t0_t1 >> 22;
  const int bx = blockIdx.x;
  const int s = (bx >= t0 ? 1 : 0) + (bx >= t0 + t1 ? 1 : 0);  // uniform
  const int ct = bx - (s >= 1 ? t0 : 0) - (s >= 2 ? t1 : 0);
  linp_body<NW, 3>(s == 0 ? A0 : (s == 1 ? A1 : A2), s == 0 ? W0 : (s == 1 ? W1 : W2), lda, K, K, B, ct, s, a, red);
}

// Single-segment link with SCALAR arguments: the command processor preloads the first 16 argument dwords into SGPRs before
// the wave starts (-mllvm -amdgpu-kernarg-preload-count=16; struct arguments are never preloaded), so the operand loads do not
// wait for a kernarg fetch from memory (0.16-0.2 us of a ~3 us link, tools/chain_bench.hip "scalar args").  Everything the
// loads need sits in the 14 dwords that are preloaded (16 user SGPRs minus the kernarg pointer): five pointers and the leading
// dimensions / K / B / flags packed two 16-bit values a dword; `out` and `slope`, needed last, come by s_load.
template <int NW>
__global__ __launch_bounds__(NW * 64) void lin1_stage_kernel(const float* A, const float* W, const float* bias,
                                                             const float* add, const float* gate, unsigned lda_ldw,
                                                             unsigned k_b, unsigned ldadd_ldgate, unsigned ldo_flags,
                                                             float* out, float slope) {
  __shared__ float red[NW * 256];
  const int lda = lda_ldw & 0xffff, ldw = lda_ldw >> 16, K = k_b & 0xffff, B = k_b >> 16;
  const int ldadd = ldadd_ldgate & 0xffff, ldgate = ldadd_ldgate >> 16, ldo = ldo_flags & 0xffff, flags = ldo_flags >> 16;
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const float e_bias = bias[(flags & LF_BIAS) ? col : 0];
  const float e_add = add[(flags & LF_ADD) ? (size_t)rowc * ldadd + col : 0];
  const float e_gate = gate[(flags & LF_GATE) ? (size_t)rowc * ldgate + col : 0];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW, true>(A, lda, r0, B, W, ldw, c0, K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (!own) return;
  float x = v[0] + ((flags & LF_BIAS) ? e_bias : 0.f) + ((flags & LF_ADD) ? e_add : 0.f);
  if (flags & LF_RELU) x = x > 0.f ? x : x * slope;
  if (flags & LF_GATE) x = e_gate > 0.f ? x : x * slope;
  out[(size_t)row * ldo + col] = x;
}

// Two SYMMETRIC segments (prior | posterior layer of the same shape, forward with a bias or backward with a gate) with scalar
// arguments: both segments share lda, K (= ldw, true for every T16 weight that is not a K-slice), ldo, the tile count and the
// flags, and have at most ONE epilogue operand each (bias or gate), so all that the loads need fits the 14 preloaded dwords:
// A0 A1 W0 W1 e0 e1 | lda/16:12 K/16:12 flags:4 | B:12 tiles:10 lde/4:10.  out0/out1, ldo, slope come by s_load.
template <int NW>
__global__ __launch_bounds__(NW * 64) void lin2s_stage_kernel(const float* A0, const float* A1, const float* W0, const float* W1,
                                                              const float* e0, const float* e1, unsigned lda_k_flags,
                                                              unsigned b_tiles_lde, float* out0, float* out1, int ldo,
                                                              float slope) {
  __shared__ float red[NW * 256];
  const int lda = (lda_k_flags & 0xfff) * 16, K = ((lda_k_flags >> 12) & 0xfff) * 16, flags = lda_k_flags >> 24;
  const int B = b_tiles_lde & 0xfff, tiles = (b_tiles_lde >> 12) & 0x3ff, lde = (b_tiles_lde >> 22) * 4;
  const bool s1 = (int)blockIdx.x >= tiles;  // uniform
  const int ct = (int)blockIdx.x - (s1 ? tiles : 0);
  const float* A = s1 ? A1 : A0;
  const float* W = s1 ? W1 : W0;
  const float* e = s1 ? e1 : e0;
  const int r0 = blockIdx.y * 16, c0 = ct * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  // bias: e[col]; gate: e[row * lde + col]; neither: e is a valid dummy
  const float e_val = e[(flags & LF_GATE) ? (size_t)rowc * lde + col : ((flags & LF_BIAS) ? col : 0)];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW, true>(A, lda, r0, B, W, K, c0, K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (!own) return;
  float x = v[0] + ((flags & LF_BIAS) ? e_val : 0.f);
  if (flags & LF_RELU) x = x > 0.f ? x : x * slope;
  if (flags & LF_GATE) x = e_val > 0.f ? x : x * slope;
  (s1 ? out1 : out0)[(size_t)row * ldo + col] = x;
}

// The same stage on 32x32 tiles for large batches (B >= 128): `tiles` counts 32-column tiles.  The NW partial tiles are
// combined through LDS (row stride 33: conflict-free); threads 0..255 then own 4 output elements each.
template <int NW, int NSEG>
__global__ __launch_bounds__(NW * 64) void lin_stage32_kernel(LinArgs<NSEG> a) {
  __shared__ float red[NW * 32 * 33];
  int ct = blockIdx.x, s = 0;
  if (NSEG > 1) {
    const int t0 = a.tiles[0], t1 = a.tiles[NSEG > 1 ? 1 : 0];
    s = (ct >= t0 ? 1 : 0) + ((NSEG > 2 && ct >= t0 + t1) ? 1 : 0);
    ct -= (s >= 1 ? t0 : 0) + (s >= 2 ? t1 : 0);
  }
  const float* A = PICK(A);
  const float* W = PICK(W);
  const float* bias = PICK(bias);
  const float* add = PICK(add);
  const float* gate = PICK(gate);
  float* out = PICK(out);
  const int lda = PICK(lda), ldw = PICK(ldw), ldadd = PICK(ldadd), ldgate = PICK(ldgate), ldo = PICK(ldo);
  const int flags = PICK(flags);
  const int K = PICK(K);
  const int r0 = blockIdx.y * 32, c0 = ct * 32;
  const int t = threadIdx.x & 255, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // epilogue operands of this thread's 4 elements (e = t + 256 q: row r0 + e/32, column c0 + e%32), before the K loop
  float e_bias[4], e_add[4], e_gate[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = t + 256 * q, row = r0 + (e >> 5), col = c0 + (e & 31);
    const int rowc = row < a.B ? row : r0;
    e_bias[q] = bias[(flags & LF_BIAS) ? col : 0];
    e_add[q] = add[(flags & LF_ADD) ? (size_t)rowc * ldadd + col : 0];
    e_gate[q] = gate[(flags & LF_GATE) ? (size_t)rowc * ldgate + col : 0];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = wave_gemm32<NW, true>(A, lda, r0, a.B, W, ldw, c0, K, wave, acc);
  {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * (32 * 33) + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
  }
  __syncthreads();
  if (threadIdx.x >= 256) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = t + 256 * q, rr = e >> 5, cc = e & 31, row = r0 + rr;
    if (row >= a.B) continue;
    float x = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) x += red[w * (32 * 33) + rr * 33 + cc];
    x += ((flags & LF_BIAS) ? e_bias[q] : 0.f) + ((flags & LF_ADD) ? e_add[q] : 0.f);
    if (flags & LF_RELU) x = x > 0.f ? x : x * a.slope;
    if (flags & LF_GATE) x = e_gate[q] > 0.f ? x : x * a.slope;
    out[(size_t)row * ldo + c0 + cc] = x;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// F4: both Gaussian heads + residual mean + reparameterised sample
// ---------------------------------------------------------------------------------------------------------------
struct HeadArgs {
  const float *P, *Q;                 // [B,H] last hidden of prior / posterior MLP
  const float *Wp, *bp, *Wq, *bq;     // [2Z,H] in T16, [2Z]
  const float* eps;                   // [B,Z]
  float *mu_p, *sd_p, *mu_q, *sd_q, *z, *raw_p, *raw_q;  // [B,Z]
  float* muq_raw;  // [B,Z] posterior mean BEFORE the combination with the prior (needed by backward in mode 2), or null
  int B, H, Z, residual;  // residual: 0 plain, 1 mu_q += mu_p, 2 precision-weighted product of q and p (variational.py:125-138),
                          // 3 GENERATION: z is drawn from the prior (q := p), `VRNNCell.generate` vrnn.py:144-163, `RSSMCell.generate`
  float beta, inv_beta, sd_eps;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void head_stage_kernel(const float* P, const float* Q, const float* Wp, const float* Wq,
                                                             const float* bp, const float* bq, unsigned b_h, int Z, HeadArgs a) {
  // the leading scalars (14 dwords: what the operand and bias loads need) are preloaded into SGPRs, the struct comes by s_load
  const int B = b_h & 0xffff, H = b_h >> 16;
  __shared__ float red[4 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const size_t o = (size_t)row * Z + col;
  const size_t oc = (size_t)(row < B ? row : r0) * Z + col;  // clamped: unconditional prefetch
  const float b0 = bp[col], b1 = bp[Z + col], b2 = bq[col], b3 = bq[Z + col];
  float e = 0.f;
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[4] = {P, P, Q, Q};
    const float* const Ws[4] = {Wp, Wp, Wq, Wq};
    const int ld[4] = {H, H, H, H}, cs[4] = {c0, Z + c0, c0, Z + c0};
    const float* eps = a.eps;
    wave_gemm16_multi<NW, 4, false>(As, ld, r0, B, Ws, ld, cs, H, wave, acc, [&]() { e = eps[oc]; });
  }
  float v[4];
  reduce_tiles<4, NW>(acc, red, v);
  if (!own) return;
  const float mp = v[0] + b0;
  const float rp = v[1] + b1;
  float mq = v[2] + b2;
  const float rq = v[3] + b3;
  const float sp = softplus_beta(rp, a.beta, a.inv_beta) + a.sd_eps;
  const float sq = softplus_beta(rq, a.beta, a.inv_beta) + a.sd_eps;
  if (a.muq_raw != nullptr) a.muq_raw[o] = mq;
  float sqc = sq;
  if (a.residual == 1) {
    mq += mp;
  } else if (a.residual == 2) {
    const float pq = 1.f / (sq * sq), pp = 1.f / (sp * sp);
    const float var = 1.f / (pq + pp);
    mq = var * (mq * pq + mp * pp);
    sqc = sqrtf(var);
  } else if (a.residual == 3) {
    mq = mp;
    sqc = sp;
  }
  a.mu_p[o] = mp; a.sd_p[o] = sp; a.mu_q[o] = mq; a.sd_q[o] = sqc;
  a.raw_p[o] = rp; a.raw_q[o] = rq;
  a.z[o] = e * sqc + mq;  // randn_like(mu).mul(sd).add(mu)
}

inline void launch_head(const HeadArgs& h, int nw, dim3 grid, hipStream_t s) {
  const unsigned b_h = (unsigned)h.B | ((unsigned)h.H << 16);  // callers require B, H < 65536
  if (nw == 16) hipLaunchKernelGGL((head_stage_kernel<16>), grid, dim3(1024), 0, s, h.P, h.Q, h.Wp, h.Wq, h.bp, h.bq, b_h, h.Z, h);
  else if (nw == 8) hipLaunchKernelGGL((head_stage_kernel<8>), grid, dim3(512), 0, s, h.P, h.Q, h.Wp, h.Wq, h.bp, h.bq, b_h, h.Z, h);
  else hipLaunchKernelGGL((head_stage_kernel<4>), grid, dim3(256), 0, s, h.P, h.Q, h.Wp, h.Wq, h.bp, h.bq, b_h, h.Z, h);
}

// ---------------------------------------------------------------------------------------------------------------
// B6: dz = dphi0 W_phi0, then through rsample / residual / KL(+free nats) / softplus heads
// ---------------------------------------------------------------------------------------------------------------
struct DzArgs {
  const float* D;    // [B,H] grad wrt the pre-activation of the layer that consumes z (VRNN: phi layer 0)
  const float* WT;   // [Z,H] that layer's weight, transposed, in T16
  const float* D2;   // optional second consumer of z (SRNN: posterior layer 0 of the NEXT step), or null
  const float* WT2;  // [Z,H]
  const float* dz_add;  // optional [B,Z] (row stride ld_add) direct gradient wrt z (SRNN: from the decoder), or null
  int ld_add, has_gemm;
  const float *mu_q, *sd_q, *mu_p, *sd_p, *eps, *raw_q, *raw_p;  // [B,Z] (step t)
  const float* muq_raw;  // [B,Z] un-combined posterior mean (mode 2 only)
  const int32_t* x_sl;
  const float *c_raw, *c_fn;  // [B] or null
  float *dqh, *dph;   // [B,2Z] grads wrt the heads' Linear outputs
  int B, H, Z, residual, t, stride;
  float fn_floor, beta, sd_eps;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void dz_stage_kernel(const float* D, const float* WT, const float* D2, const float* WT2,
                                                           unsigned b_h, int Z, int has_gemm, DzArgs a) {
  // the leading scalars (what the operand loads need) are preloaded into SGPRs; the struct comes by s_load, and every saved
  // value of the step is prefetched by `mid`, after the operand loads have been issued
  const int B = b_h & 0xffff, H = b_h >> 16;
  __shared__ float red[2 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;  // clamped: unconditional prefetch
  const size_t o = (size_t)rowc * Z + col;
  float mq = 0.f, sq = 1.f, mp = 0.f, sp = 1.f, e = 0.f, rq = 0.f, rp = 0.f, c_raw = 0.f, c_fn = 0.f, e_add = 0.f;
  auto prefetch = [&]() {
    mq = a.mu_q[o]; sq = a.sd_q[o]; mp = a.mu_p[o]; sp = a.sd_p[o]; e = a.eps[o]; rq = a.raw_q[o]; rp = a.raw_p[o];
    if (a.c_fn != nullptr || a.c_raw != nullptr) {  // wave-uniform
      const bool live = (long long)a.t * a.stride < a.x_sl[rowc];
      const float cr = a.c_raw != nullptr ? a.c_raw[rowc] : 0.f;
      const float cf = a.c_fn != nullptr ? a.c_fn[rowc] : 0.f;
      c_raw = live ? cr : 0.f;
      c_fn = live ? cf : 0.f;
    }
    e_add = a.dz_add != nullptr ? a.dz_add[(size_t)rowc * a.ld_add + col] : 0.f;
  };
  float v[2] = {0.f, 0.f};
  if (has_gemm) {  // wave-uniform
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (D2 != nullptr) {  // wave-uniform
      const float* const As[2] = {D, D2};
      const float* const Ws[2] = {WT, WT2};
      const int ld[2] = {H, H}, cs[2] = {c0, c0};
      wave_gemm16_multi<NW, 2, false>(As, ld, r0, B, Ws, ld, cs, H, threadIdx.x >> 6, acc, prefetch);
    } else {
      acc[0] = wave_gemm16<NW, true>(D, H, r0, B, WT, H, c0, H, threadIdx.x >> 6, acc[0], prefetch);
    }
    reduce_tiles<2, NW>(acc, red, v);
  } else {
    prefetch();
  }
  if (!own) return;
  const float dz = v[0] + v[1] + e_add;
  const float d = mq - mp, ip2 = 1.f / (sp * sp);
  float coef = c_raw;
  if (c_fn != 0.f) {
    const float k = logf(sp) - logf(sq) + (sq * sq + d * d) * 0.5f * ip2 - 0.5f;
    if (!(a.fn_floor > 0.f) || k > a.fn_floor) coef += c_fn;
  }
  // gradients wrt the (combined) posterior (mq, sq) and the prior (mp, sp) of: rsample + KL(+free nats)
  float g_muq = dz + coef * d * ip2;
  float g_sdq = dz * e + coef * (sq * ip2 - 1.f / sq);
  float g_mup = -coef * d * ip2;
  float g_sdp = coef * (1.f / sp - (sq * sq + d * d) * ip2 / sp);
  if (a.residual == 1) {
    g_mup += g_muq;  // mu_q = mu_q' + mu_p
  } else if (a.residual == 2) {
    // (mq, sq) = precision-weighted product of q' = (muq_raw, softplus(raw_q)+eps) and p: var = 1/(pq+pp),
    // mq = var (mq' pq + mp pp), sq = sqrt(var)
    const float mqr = a.muq_raw[o];
    const float sqr = softplus_beta(rq, a.beta, 1.f / a.beta) + a.sd_eps;
    const float pq = 1.f / (sqr * sqr), pp = ip2, var = sq * sq;
    const float half_s3 = 0.5f * var * sq;
    const float g_pq = g_muq * var * (mqr - mq) - g_sdq * half_s3;
    const float g_pp = g_muq * var * (mp - mq) - g_sdq * half_s3;
    g_mup += g_muq * var * pp;
    g_sdp += g_pp * (-2.f * pp / sp);
    g_sdq = g_pq * (-2.f * pq / sqr);
    g_muq = g_muq * var * pq;
  }
  const size_t o2 = (size_t)row * 2 * Z + col;
  a.dqh[o2] = g_muq;
  a.dqh[o2 + Z] = g_sdq * sigmoidf_(a.beta * rq);
  a.dph[o2] = g_mup;
  a.dph[o2 + Z] = g_sdp * sigmoidf_(a.beta * rp);
}

inline void launch_dz(const DzArgs& d, int nw, dim3 grid, hipStream_t s) {
  const unsigned b_h = (unsigned)d.B | ((unsigned)d.H << 16);  // callers require B, H < 65536
  if (nw == 16) hipLaunchKernelGGL((dz_stage_kernel<16>), grid, dim3(1024), 0, s, d.D, d.WT, d.D2, d.WT2, b_h, d.Z, d.has_gemm, d);
  else if (nw == 8) hipLaunchKernelGGL((dz_stage_kernel<8>), grid, dim3(512), 0, s, d.D, d.WT, d.D2, d.WT2, b_h, d.Z, d.has_gemm, d);
  else hipLaunchKernelGGL((dz_stage_kernel<4>), grid, dim3(256), 0, s, d.D, d.WT, d.D2, d.WT2, b_h, d.Z, d.has_gemm, d);
}

// number of waves for a K-deep reduction.  One product (groups == 1, wave_gemm16): a wave's chunks cost one memory round trip
// per 4 chunks plus one per left-over chunk, so take the NW with the fewest trips (K = 192: 12 chunks -> 16 waves x 1 chunk, not
// 4 waves x 3), the size rule's choice on ties.  Several products (wave_gemm16_multi): <= 4 chunks of 16 per wave over all products.
inline int pick_nw(int K, int groups) {
  const int chunks = (K / 16) * groups;
  const int by_size = chunks > 32 ? 16 : (chunks > 16 ? 8 : 4);  // <= 4 chunks per wave where possible
  if (groups != 1) return by_size;
  auto trips = [K](int nw) { const int cpw = (K / 16 + nw - 1) / nw; return cpw / 4 + cpw % 4; };
  int best = by_size;
  for (int nw = 4; nw <= 16; nw *= 2)
    if (trips(nw) < trips(best)) best = nw;
  return best;
}

#define LAUNCH_NW(kernel, nw, grid, stream, args)                                          \
  do {                                                                                     \
    if ((nw) == 16) hipLaunchKernelGGL((kernel<16>), grid, dim3(1024), 0, stream, args);   \
    else if ((nw) == 8) hipLaunchKernelGGL((kernel<8>), grid, dim3(512), 0, stream, args); \
    else hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, stream, args);                \
  } while (0)


struct LinSegH {  // host-side description of one segment
  const float* A; int lda; const float* W; int ldw; const float* bias; const float* add; int ldadd;
  const float* gate; int ldgate; float* out; int ldo; int ncols, K, relu;
};

inline LinSegH seg(const float* A, int lda, const float* W, int ldw, const float* bias, const float* add, int ldadd,
                   const float* gate, int ldgate, float* out, int ldo, int ncols, int K, int relu) {
  return LinSegH{A, lda, W, ldw, bias, add, ldadd, gate, ldgate, out, ldo, ncols, K, relu};
}

struct LinLaunch {
  LinSegH seg[3];
  int nseg, B;
  float slope = 0.f;
};

// large batches: 32x32 tiles when every segment's width allows it (env BLVM_LIN32_MIN_B overrides the threshold, 0 = never)
inline int lin32_min_batch() {
  static int v = [] {
    const char* e = getenv("BLVM_LIN32_MIN_B");
    return e ? atoi(e) : 128;
  }();
  return v;
}

template <int NSEG>
inline void launch_lin_n(const LinLaunch& l, hipStream_t s) {
  LinArgs<NSEG> a{};
  int tiles = 0, kmax = 0;
  bool wide = lin32_min_batch() > 0 && l.B >= lin32_min_batch();
  for (int i = 0; i < NSEG; ++i) wide = wide && (l.seg[i].ncols % 32 == 0) && (l.seg[i].K % 8 == 0);
  const int tw = wide ? 32 : 16;
  for (int i = 0; i < NSEG; ++i) {
    const LinSegH& g = l.seg[i];
    a.A[i] = g.A; a.W[i] = g.W; a.out[i] = g.out;
    a.bias[i] = g.bias ? g.bias : g.W;   // valid dummy for absent operands
    a.add[i] = g.add ? g.add : g.W;
    a.gate[i] = g.gate ? g.gate : g.W;
    a.lda[i] = g.lda; a.ldw[i] = g.ldw; a.ldadd[i] = g.ldadd; a.ldgate[i] = g.ldgate; a.ldo[i] = g.ldo;
    a.tiles[i] = g.ncols / tw;
    a.flags[i] = (g.bias ? LF_BIAS : 0) | (g.add ? LF_ADD : 0) | (g.gate ? LF_GATE : 0) | (g.relu ? LF_RELU : 0);
    a.K[i] = g.K;
    kmax = g.K > kmax ? g.K : kmax;
    tiles += g.ncols / tw;
  }
  a.B = l.B;
  a.slope = l.slope;
  if (wide) {  // chunks of 8 k: <= 8 chunks per wave where possible (a wave's dependent chain is 4 MFMAs of 64 cycles per chunk)
    const int chunks = kmax / 8;
    const int nw32 = chunks > 64 ? 16 : (chunks > 32 ? 8 : 4);
    const dim3 grid32(tiles, (l.B + 31) / 32);
    if (nw32 == 16) hipLaunchKernelGGL((lin_stage32_kernel<16, NSEG>), grid32, dim3(1024), 0, s, a);
    else if (nw32 == 8) hipLaunchKernelGGL((lin_stage32_kernel<8, NSEG>), grid32, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((lin_stage32_kernel<4, NSEG>), grid32, dim3(256), 0, s, a);
    return;
  }
  const int nw = pick_nw(kmax, 1);
  const dim3 grid(tiles, (l.B + 15) / 16);
#ifndef BLVM_NO_LIN1
  if (NSEG == 1 && l.B < 65536 && a.K[0] < 65536 && a.lda[0] < 65536 && a.ldw[0] < 65536 && a.ldadd[0] < 65536 &&
      a.ldgate[0] < 65536 && a.ldo[0] < 65536) {
    const unsigned p0 = (unsigned)a.lda[0] | ((unsigned)a.ldw[0] << 16), p1 = (unsigned)a.K[0] | ((unsigned)l.B << 16);
    const unsigned p2 = (unsigned)a.ldadd[0] | ((unsigned)a.ldgate[0] << 16), p3 = (unsigned)a.ldo[0] | ((unsigned)a.flags[0] << 16);
    if (nw == 16) hipLaunchKernelGGL((lin1_stage_kernel<16>), grid, dim3(1024), 0, s, a.A[0], a.W[0], a.bias[0], a.add[0], a.gate[0], p0, p1, p2, p3, a.out[0], a.slope);
    else if (nw == 8) hipLaunchKernelGGL((lin1_stage_kernel<8>), grid, dim3(512), 0, s, a.A[0], a.W[0], a.bias[0], a.add[0], a.gate[0], p0, p1, p2, p3, a.out[0], a.slope);
    else hipLaunchKernelGGL((lin1_stage_kernel<4>), grid, dim3(256), 0, s, a.A[0], a.W[0], a.bias[0], a.add[0], a.gate[0], p0, p1, p2, p3, a.out[0], a.slope);
    return;
  }
#endif
#ifndef BLVM_NO_LIN2S
  if (NSEG == 2) {
    const int f = a.flags[0];
    const bool one_operand = (f & LF_ADD) == 0 && ((f & LF_BIAS) == 0 || (f & LF_GATE) == 0);
    const int lde = (f & LF_GATE) ? a.ldgate[0] : 0;
    if (one_operand && a.flags[1 % NSEG] == f && a.lda[0] == a.lda[1 % NSEG] && a.K[0] == a.K[1 % NSEG] && a.ldw[0] == a.K[0] &&
        a.ldw[1 % NSEG] == a.K[0] && a.ldo[0] == a.ldo[1 % NSEG] && a.tiles[0] == a.tiles[1 % NSEG] &&
        (!(f & LF_GATE) || a.ldgate[1 % NSEG] == lde) && l.B < 4096 && a.lda[0] % 16 == 0 && a.lda[0] < 65536 &&
        a.K[0] % 16 == 0 && a.K[0] < 65536 && lde % 4 == 0 && lde < 4096 && a.tiles[0] < 1024) {
      const float* e0 = (f & LF_GATE) ? a.gate[0] : a.bias[0];
      const float* e1 = (f & LF_GATE) ? a.gate[1 % NSEG] : a.bias[1 % NSEG];
      const unsigned p0 = (unsigned)(a.lda[0] / 16) | ((unsigned)(a.K[0] / 16) << 12) | ((unsigned)f << 24);
      const unsigned p1 = (unsigned)l.B | ((unsigned)a.tiles[0] << 12) | ((unsigned)(lde / 4) << 22);
      const int p2 = a.ldo[0];
      if (nw == 16) hipLaunchKernelGGL((lin2s_stage_kernel<16>), grid, dim3(1024), 0, s, a.A[0], a.A[1 % NSEG], a.W[0], a.W[1 % NSEG], e0, e1, p0, p1, a.out[0], a.out[1 % NSEG], p2, a.slope);
      else if (nw == 8) hipLaunchKernelGGL((lin2s_stage_kernel<8>), grid, dim3(512), 0, s, a.A[0], a.A[1 % NSEG], a.W[0], a.W[1 % NSEG], e0, e1, p0, p1, a.out[0], a.out[1 % NSEG], p2, a.slope);
      else hipLaunchKernelGGL((lin2s_stage_kernel<4>), grid, dim3(256), 0, s, a.A[0], a.A[1 % NSEG], a.W[0], a.W[1 % NSEG], e0, e1, p0, p1, a.out[0], a.out[1 % NSEG], p2, a.slope);
      return;
    }
  }
#endif
#ifndef BLVM_NO_LINP
  if (NSEG == 2) {
    bool fits = l.B < 65536;
    for (int i = 0; i < 2; ++i) fits = fits && a.lda[i % NSEG] < 65536 && a.ldw[i % NSEG] < 65536 && a.K[i % NSEG] < 65536;
    fits = fits && a.tiles[0] < 65536;
    if (fits) {
      const LinArgs<2>& a2 = reinterpret_cast<const LinArgs<2>&>(a);  // NSEG == 2 here
      const unsigned p0 = (unsigned)a2.lda[0] | ((unsigned)a2.lda[1] << 16), p1 = (unsigned)a2.ldw[0] | ((unsigned)a2.ldw[1] << 16);
      const unsigned p2 = (unsigned)a2.K[0] | ((unsigned)a2.K[1] << 16), p3 = (unsigned)a2.tiles[0] | ((unsigned)l.B << 16);
      if (nw == 16) hipLaunchKernelGGL((linp2_stage_kernel<16>), grid, dim3(1024), 0, s, a2.A[0], a2.A[1], a2.W[0], a2.W[1], p0, p1, p2, p3, a2);
      else if (nw == 8) hipLaunchKernelGGL((linp2_stage_kernel<8>), grid, dim3(512), 0, s, a2.A[0], a2.A[1], a2.W[0], a2.W[1], p0, p1, p2, p3, a2);
      else hipLaunchKernelGGL((linp2_stage_kernel<4>), grid, dim3(256), 0, s, a2.A[0], a2.A[1], a2.W[0], a2.W[1], p0, p1, p2, p3, a2);
      return;
    }
  }
  if (NSEG == 3) {
    const LinArgs<3>& a3 = reinterpret_cast<const LinArgs<3>&>(a);  // NSEG == 3 here
    bool fits = l.B < 4096 && a3.tiles[0] < 1024 && a3.tiles[1] < 1024 && a3.lda[0] < 65536 && a3.K[0] < 65536;
    for (int i = 0; i < 3; ++i) fits = fits && a3.lda[i] == a3.lda[0] && a3.K[i] == a3.K[0] && a3.ldw[i] == a3.K[0];
    if (fits) {
      const unsigned p0 = (unsigned)a3.lda[0] | ((unsigned)a3.K[0] << 16);
      const unsigned p1 = (unsigned)l.B | ((unsigned)a3.tiles[0] << 12) | ((unsigned)a3.tiles[1] << 22);
      if (nw == 16) hipLaunchKernelGGL((linp3_stage_kernel<16>), grid, dim3(1024), 0, s, a3.A[0], a3.A[1], a3.A[2], a3.W[0], a3.W[1], a3.W[2], p0, p1, a3);
      else if (nw == 8) hipLaunchKernelGGL((linp3_stage_kernel<8>), grid, dim3(512), 0, s, a3.A[0], a3.A[1], a3.A[2], a3.W[0], a3.W[1], a3.W[2], p0, p1, a3);
      else hipLaunchKernelGGL((linp3_stage_kernel<4>), grid, dim3(256), 0, s, a3.A[0], a3.A[1], a3.A[2], a3.W[0], a3.W[1], a3.W[2], p0, p1, a3);
      return;
    }
  }
#endif
  if (nw == 16) hipLaunchKernelGGL((lin_stage_kernel<16, NSEG>), grid, dim3(1024), 0, s, a);
  else if (nw == 8) hipLaunchKernelGGL((lin_stage_kernel<8, NSEG>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((lin_stage_kernel<4, NSEG>), grid, dim3(256), 0, s, a);
}

inline void launch_lin(const LinLaunch& l, hipStream_t s) {
  if (l.nseg == 1) launch_lin_n<1>(l, s);
  else if (l.nseg == 2) launch_lin_n<2>(l, s);
  else launch_lin_n<3>(l, s);
}

inline int pick_split(int M, int N, int K) { return gemm_pick_split(M, N, K); }

// dW (+)= D^T Act over all rows; db (+)= column sums of D in the same launch (null: none)
inline int bgrad(const float* D, int ldd, int n_out, float* db, size_t rows, hipStream_t s);
inline int wgrad(const float* D, int ldd, int n_out, const float* Act, int lda, int k_in, float* dW, int ldw, size_t rows,
          hipStream_t s, float* db = nullptr) {
  if (!dW) return bgrad(D, ldd, n_out, db, rows, s);
  return gemm_f32(1, 1, n_out, k_in, (int)rows, D, ldd, Act, lda, dW, ldw, nullptr, 0, 0.f, nullptr, 0, 1,
                  pick_split(n_out, k_in, (int)rows), s, db);
}

// the weight gradients of one sequence collected and run as ONE grouped launch (gemm.hip gemm_wgrad_group); add() takes wgrad()'s
// arguments without the row count and the stream
struct WgradGroup {
  std::vector<WgradJob> jobs;
  void add(const float* D, int ldd, int n_out, const float* Act, int lda, int k_in, float* dW, int ldw, float* db = nullptr) {
    jobs.push_back(WgradJob{D, ldd, n_out, Act, lda, k_in, dW, ldw, db});
  }
  int run(size_t rows, hipStream_t s) { return gemm_wgrad_group(jobs.data(), (int)jobs.size(), (int)rows, s); }
};

inline int bgrad(const float* D, int ldd, int n_out, float* db, size_t rows, hipStream_t s) {
  if (!db) return BLVM_OK;
  return colsum_f32((int)rows, n_out, D, ldd, db, 1, s);
}

