// pchain_rt.h — ROW-GROUP tiles of the persistent-chain engine (pchain.h) for batches beyond 128 utterances.
//
// pchain.h's tile is 16 rows x 16 columns: what a tile costs (descriptor walk, weight fetch, one hand-off round trip, K-split
// reduction, epilogue: ~2.4 us) does not depend on how many rows it carries, and at B = 256 a link is 256 such tiles — four per
// workgroup, back to back, ~9.6 us per link: slower than a launch per link on 32 x 32 tiles.  Here a tile is a GROUP of up to RT
// row tiles of one column tile (RT x 16 rows x 16 columns): the weight fragments of a k-chunk are fetched ONCE and multiply all RT
// activation fragments (whose polled loads travel together: one round trip for the group), the accumulators are reduced and
// finished row tile by row tile through the same LDS scratch.  The T16 operand layout, the sentinel hand-off, the descriptors and
// every epilogue are pchain.h's (the epilogue bodies below restate them line by line; results agree to fp32 summation order:
// same chunk order per wave, same cross-wave sum).  Only the tile kinds of the VRNN programs exist in this form.
#pragma once
#include "pchain.h"

namespace blvm {
namespace pchain {

#ifndef BLVM_RT_FRAGS
#define BLVM_RT_FRAGS 16
#endif
constexpr int kRtFrags = BLVM_RT_FRAGS;

// acc[rt][g] += A[AMap(g)][r0 + 16 rt + i][k] W[g][c0[g] + j][k] for the row tiles rt < RT of a group (mgemm_trip of pchain.h with
// the activation side repeated per row tile).  rt_off[rt]: byte offset of row tile rt's T16 slab from the group's first (clamped to
// the last row tile that exists, so no load leaves the step's slab; aok[rt] is false for every lane of a row tile beyond B).
template <int NW, bool BF, int RT, int GA, int G, class AMap, int CH, class Mid>
__device__ __forceinline__ void mgemm_trip_rt(const rsrc_t (&ar)[GA], unsigned aoff, const unsigned (&rt_off)[RT], const float* const (&ap)[GA],
                                              const size_t (&rt_rows)[RT][GA], const char* const (&wp)[G], int kc, const bool (&aok)[RT], bool polled,
                                              f32x4 (&acc)[RT][G], Poll& pl, Mid& mid, bool& mid_pending) {
  constexpr int STEP = NW * 16;
  typedef typename WFrag<BF>::type wfrag;
  constexpr int ES = BF ? 2 : 4;
  wfrag w[G][CH];
  f32x4 a[RT][GA][CH];
#pragma unroll
  for (int u = 0; u < CH; ++u)
#pragma unroll
    for (int g = 0; g < G; ++g) w[g][u] = *reinterpret_cast<const wfrag*>(wp[g] + (size_t)ES * 16 * (size_t)(kc + u * STEP));
  if (!polled) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int g = 0; g < GA; ++g) a[rt][g][u] = *reinterpret_cast<const f32x4*>(ap[g] + rt_rows[rt][g] + kc + u * STEP);
    if (mid_pending) { mid(); mid_pending = false; }
  } else {
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
          for (int g = 0; g < GA; ++g) a[rt][g][u] = ld_sc1_x4(ar[g], aoff + rt_off[rt] + 64u * (unsigned)(kc + u * STEP));
      if (mid_pending) { mid(); mid_pending = false; }
      bool bad = false;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        bool b = false;
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
          for (int g = 0; g < GA; ++g) b |= any_sentinel(a[rt][g][u]);
        bad |= b && aok[rt];
      }
      if (!__any(bad) || pl.dead) break;
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
  }
  if constexpr (AMap::sum) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int ga = 1; ga < GA; ++ga) a[rt][0][u] += a[rt][ga][u];
  }
  if constexpr (BF) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        s16x4 ab[GA];
#pragma unroll
        for (int ga = 0; ga < GA; ++ga) {
          const f32x4 x = a[rt][ga][u];
          const u32x2 q = {aok[rt] ? pk_bf16(x[0], x[1]) : 0u, aok[rt] ? pk_bf16(x[2], x[3]) : 0u};
          ab[ga] = __builtin_bit_cast(s16x4, q);
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
          acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[AMap::of(g)], __builtin_bit_cast(s16x4, w[g][u]), acc[rt][g], 0, 0, 0);
      }
  } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < G; ++g)
            acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aok[rt] ? a[rt][AMap::of(g)][u][e] : 0.f, w[g][u][e], acc[rt][g], 0, 0, 0);
  }
}

template <int NW, bool BF, int RT, int GA, int G, class AMap, class Mid = NoMid>
__device__ __forceinline__ void mgemm16_rt(const float* const (&A)[GA], const int (&lda)[GA], bool polled, int r0, int nrows, const float* const (&W)[G],
                                           const int (&c0)[G], int K, f32x4 (&acc)[RT][G], Poll& pl, Mid mid = Mid(), int a_width = 0, int w_width = 0) {
  constexpr int STEP = NW * 16;
  constexpr int FR = kRtFrags / (G + GA * RT);  // fragments in flight per trip (512 threads: 256 VGPRs a lane)
  constexpr int MAXCH = FR >= 6 ? 6 : (FR >= 4 ? 4 : (FR >= 2 ? 2 : 1));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane & 15, q = lane >> 4;
  const int width = a_width > 0 ? a_width : K;
  const int last_rt = (nrows - 1 - r0) >> 4;  // last row tile of the group that exists (>= 0: the group's first always does)
  bool aok[RT];
  unsigned rt_off[RT];
  size_t rt_rows[RT][GA];
  rsrc_t ar[GA];
  const float* ap[GA];
  const char* wp[G];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    aok[rt] = (r0 + 16 * rt + rr) < nrows;
    const int rtc = rt < last_rt ? rt : last_rt;
    rt_off[rt] = 64u * (unsigned)rtc * (unsigned)width;  // a row tile's T16 slab is 16 * width floats
#pragma unroll
    for (int g = 0; g < GA; ++g) rt_rows[rt][g] = (size_t)(aok[rt] ? r0 + 16 * rt + rr : r0) * (polled ? 0 : lda[g]);
  }
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    ar[g] = make_rsrc(A[g]);
    ap[g] = A[g] + 4 * q;
  }
  const unsigned aoff = 4u * ((unsigned)(r0 >> 4) * 16u * (unsigned)width + 4u * (unsigned)lane);
#pragma unroll
  for (int g = 0; g < G; ++g) wp[g] = reinterpret_cast<const char*>(W[g]) + (BF ? 2 : 4) * ((size_t)c0[g] * (w_width > 0 ? w_width : K) + 4 * lane);
  int nch = (K / 16 - wave + NW - 1) / NW;
  int kc = wave * 16;
  bool mid_pending = true;
  if constexpr (MAXCH >= 6) for (; nch >= 6; nch -= 6, kc += 6 * STEP) mgemm_trip_rt<NW, BF, RT, GA, G, AMap, 6>(ar, aoff, rt_off, ap, rt_rows, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if constexpr (MAXCH >= 4) for (; nch >= 4; nch -= 4, kc += 4 * STEP) mgemm_trip_rt<NW, BF, RT, GA, G, AMap, 4>(ar, aoff, rt_off, ap, rt_rows, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if constexpr (MAXCH >= 2) for (; nch >= 2; nch -= 2, kc += 2 * STEP) mgemm_trip_rt<NW, BF, RT, GA, G, AMap, 2>(ar, aoff, rt_off, ap, rt_rows, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  for (; nch >= 1; nch -= 1, kc += STEP) mgemm_trip_rt<NW, BF, RT, GA, G, AMap, 1>(ar, aoff, rt_off, ap, rt_rows, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if (mid_pending) mid();
}

// ---- link tiles on row groups ---------------------------------------------------------------------------------------------
// Called by ALL 512 threads (NW = 8).  The accumulators are reduced and finished TWO row tiles per workgroup barrier: threads
// 0..255 finish row tile 2 p, threads 256..511 row tile 2 p + 1 of pair p (both halves prefetched every row tile's epilogue
// operands: thread t and thread 256 + t hold the same values).  `red()` hands out the two LDS scratch buffers alternately (each
// 2 G x NW x 256 floats: the host sizes the scratch for 8 products).  (t >> 4, t & 15) = (row, column) inside a row tile.
template <int G, int NW>
__device__ __forceinline__ void reduce_pair(const f32x4 (&acc0)[G], const f32x4 (&acc1)[G], float* __restrict__ red, float (&out)[G]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = ((lane >> 4) * 4 + r) * 16 + (lane & 15);
      red[(g * NW + wave) * 256 + idx] = acc0[g][r];
      red[((G + g) * NW + wave) * 256 + idx] = acc1[g][r];
    }
  }
  __syncthreads();
  const int pm = tid >> 8, idx = tid & 255;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[((pm * G + g) * NW + w) * 256 + idx];
    out[g] = s;
  }
}

// CT: adjacent column tiles per tile (CT products of the same activation fragments against the weight rows c0 + 16 c ...).  Only
// CT = 1 is instantiated: measured (r03, B = 128 / 192 / 256, same box) a 32-column tile costs twice a 16-column one -- the fp32
// MFMAs of a tile are ~0.85 us of its ~3.1 us on a CU whose two waves per SIMD share the matrix pipe, the rest scales with the
// outputs too -- so pairs only halve the tiles a link can be spread over (B = 128: 22.7 -> 28.6 ms/step).
template <int NW, bool BF, int RT, int CT, class Late, class Red>
__device__ __forceinline__ void tile_lin_rt(const float* A, int lda, bool a_polled, const float* W, int K, Late& late, int r0, int c0, int B, Red& red,
                                            Poll& pl, const float* A2 = nullptr, const float* A3 = nullptr, int w_width = 0) {
  const int t = threadIdx.x & 255;
  const int col = c0 + (t & 15);
  float e_bias[CT], e_gate[RT][CT], e_add[RT][CT];
  int rowc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = r0 + 16 * rt + (t >> 4);
    rowc[rt] = row < B ? row : r0;
#pragma unroll
    for (int c = 0; c < CT; ++c) { e_gate[rt][c] = 1.f; e_add[rt][c] = 0.f; }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) e_bias[c] = 0.f;
  LinLate L;
  auto prefetch = [&]() {
    L = late();
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (L.bias) e_bias[c] = L.bias[col + 16 * c];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (L.gate) e_gate[rt][c] = L.gate[(size_t)rowc[rt] * L.ldgate + col + 16 * c];
        if (L.add && !L.add_polled) e_add[rt][c] = L.add[(size_t)rowc[rt] * L.ldadd + col + 16 * c];
      }
    }
  };
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* Ws[CT];
  int cs[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { Ws[c] = W; cs[c] = c0 + 16 * c; }
  if (A2 != nullptr) {  // (uniform) the operand arrives as three partial-sum slabs
    const float* const As[3] = {A, A2, A3};
    const int la[3] = {0, 0, 0};
    mgemm16_rt<NW, BF, RT, 3, CT, MapSum>(As, la, true, r0, B, Ws, cs, K, acc, pl, prefetch, lda, w_width);
  } else {
    const float* const As[1] = {A};
    const int la[1] = {lda};
    mgemm16_rt<NW, BF, RT, 1, CT, MapSame>(As, la, a_polled, r0, B, Ws, cs, K, acc, pl, prefetch, a_polled ? lda : 0, w_width);
  }
  static_assert(RT % 2 == 0 && NW == 8, "row groups are finished in pairs by 512 threads");
#pragma unroll
  for (int pr = 0; pr < RT / 2; ++pr) {
    if (r0 + 32 * pr >= B) break;  // (uniform)
    float v[CT];
    reduce_pair<CT, NW>(acc[2 * pr], acc[2 * pr + 1], red(), v);
    const bool hi = (threadIdx.x >> 8) != 0;
    const int rt = 2 * pr + (int)hi, rb = r0 + 16 * rt;
    const int row = rb + (t >> 4);
    const bool own = row < B;
    const int rc = hi ? rowc[2 * pr + 1] : rowc[2 * pr];
    float add[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) add[c] = hi ? e_add[2 * pr + 1][c] : e_add[2 * pr][c];
    if (L.add && L.add_polled) {
      rsrc_t rs[CT];
      unsigned os[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) { rs[c] = make_rsrc(L.add); os[c] = 4u * ((unsigned)rc * (unsigned)L.ldadd + (unsigned)(col + 16 * c)); }
      poll_words<CT>(rs, os, add, own, pl);
    }
    if (own) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float gate_v = hi ? e_gate[2 * pr + 1][c] : e_gate[2 * pr][c];
        float x = v[c] + e_bias[c] + add[c];
        if (L.relu) x = x > 0.f ? x : x * L.slope;
        if (L.gate) x = gate_v > 0.f ? x : x * L.slope;
        put(L.out, rb, c0 + 16 * c, row, col + 16 * c, x);
      }
    }
  }
}

template <int NW, bool BF, int RT, class Red>
__device__ __forceinline__ void tile_head_rt(const float* P, const float* Q, bool polled, const float* Wp, const float* bp, const float* Wq, const float* bq,
                                             const float* eps, const HeadOut& o, int H, int Z, int residual, float beta, float inv_beta, float sd_eps,
                                             int r0, int c0, int B, Red& red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int col = c0 + (t & 15);
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, e[RT];
  size_t oc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = r0 + 16 * rt + (t >> 4);
    oc[rt] = (size_t)(row < B ? row : r0) * Z + col;
    e[rt] = 0.f;
  }
  auto prefetch = [&]() {
    b0 = bp[col]; b1 = bp[Z + col]; b2 = bq[col]; b3 = bq[Z + col];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) e[rt] = eps[oc[rt]];
  };
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[rt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[2] = {P, Q};
    const float* const Ws[4] = {Wp, Wp, Wq, Wq};
    const int la[2] = {H, H}, cs[4] = {c0, Z + c0, c0, Z + c0};
    mgemm16_rt<NW, BF, RT, 2, 4, MapPairs>(As, la, polled, r0, B, Ws, cs, H, acc, pl, prefetch);
  }
#pragma unroll
  for (int pr = 0; pr < RT / 2; ++pr) {
    if (r0 + 32 * pr >= B) break;
    float v[4];
    reduce_pair<4, NW>(acc[2 * pr], acc[2 * pr + 1], red(), v);
    const int rt = 2 * pr + (int)(threadIdx.x >> 8), rb = r0 + 16 * rt;
    const int row = rb + (t >> 4);
    const float ee = (threadIdx.x >> 8) ? e[2 * pr + 1] : e[2 * pr];
    if (row < B) {
      const size_t oo = (size_t)row * Z + col;
      const float mp = v[0] + b0, rp = v[1] + b1, rq = v[3] + b3;
      float mq = v[2] + b2;
      const float sp = softplus_beta(rp, beta, inv_beta) + sd_eps;
      const float sq = softplus_beta(rq, beta, inv_beta) + sd_eps;
      if (o.muq_raw != nullptr) o.muq_raw[oo] = mq;
      float sqc = sq;
      if (residual == 1) {
        mq += mp;
      } else if (residual == 2) {
        const float pq = 1.f / (sq * sq), pp = 1.f / (sp * sp);
        const float var = 1.f / (pq + pp);
        mq = var * (mq * pq + mp * pp);
        sqc = sqrtf(var);
      } else if (residual == 3) {
        mq = mp;
        sqc = sp;
      }
      put(o.z, rb, c0, row, col, ee * sqc + mq);
      o.mu_p[oo] = mp; o.sd_p[oo] = sp; o.mu_q[oo] = mq; o.sd_q[oo] = sqc;
      o.raw_p[oo] = rp; o.raw_q[oo] = rq;
    }
  }
}

template <int NW, bool BF, int RT, class Red>
__device__ __forceinline__ void tile_gru_rt(const float* X, int ldx, bool polled, const float* Wih, int K, const float* xg, const float* bih, const float* gh,
                                            const float* hprev, int ldh, int R, const Out& hnew, float* rg, float* ug, float* ng, int r0, int c0, int B,
                                            Red& red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int col = c0 + (t & 15);
  const rsrc_t rgh = make_rsrc(gh), rhp = make_rsrc(hprev);
  const rsrc_t ps[4] = {rgh, rgh, rgh, rhp};
  unsigned po[RT][4];
  size_t o3[RT];
  float w[RT][4], x0[RT], x1[RT], x2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = r0 + 16 * rt + (t >> 4);
    const int rowc = row < B ? row : r0;
    o3[rt] = (size_t)rowc * 3 * R + col;
    po[rt][0] = 4u * (unsigned)o3[rt]; po[rt][1] = 4u * (unsigned)(o3[rt] + R); po[rt][2] = 4u * (unsigned)(o3[rt] + 2 * R);
    po[rt][3] = 4u * ((unsigned)rowc * (unsigned)ldh + (unsigned)col);
    w[rt][0] = w[rt][1] = w[rt][2] = w[rt][3] = 0.f;
    x0[rt] = x1[rt] = x2[rt] = 0.f;
  }
  auto prefetch = [&]() {
    const float bi0 = bih ? bih[col] : 0.f, bi1 = bih ? bih[R + col] : 0.f, bi2 = bih ? bih[2 * R + col] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w[rt][i] = ld_sc1(ps[i], po[rt][i]);
      x0[rt] = (xg ? xg[o3[rt]] : 0.f) + bi0;
      x1[rt] = (xg ? xg[o3[rt] + R] : 0.f) + bi1;
      x2[rt] = (xg ? xg[o3[rt] + 2 * R] : 0.f) + bi2;
    }
  };
  f32x4 acc[RT][3];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[rt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[1] = {X};
    const float* const Ws[3] = {Wih, Wih, Wih};
    const int la[1] = {ldx}, cs[3] = {c0, R + c0, 2 * R + c0};
    mgemm16_rt<NW, BF, RT, 1, 3, MapSame>(As, la, polled, r0, B, Ws, cs, K, acc, pl, prefetch);
  }
#pragma unroll
  for (int pr = 0; pr < RT / 2; ++pr) {
    if (r0 + 32 * pr >= B) break;
    float v[3];
    reduce_pair<3, NW>(acc[2 * pr], acc[2 * pr + 1], red(), v);
    const int hi = (int)(threadIdx.x >> 8), rt = 2 * pr + hi, rb = r0 + 16 * rt;
    const int row = rb + (t >> 4);
    const bool own = row < B;
    float ww[4];
    unsigned pp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ww[i] = hi ? w[2 * pr + 1][i] : w[2 * pr][i]; pp[i] = hi ? po[2 * pr + 1][i] : po[2 * pr][i]; }
    const float xx0 = hi ? x0[2 * pr + 1] : x0[2 * pr], xx1 = hi ? x1[2 * pr + 1] : x1[2 * pr], xx2 = hi ? x2[2 * pr + 1] : x2[2 * pr];
    if (__any(own && (is_sentinel(ww[0]) | is_sentinel(ww[1]) | is_sentinel(ww[2]) | is_sentinel(ww[3])))) poll_words<4>(ps, pp, ww, own, pl);
    if (own) {
      const float r = sigmoidf_(v[0] + xx0 + ww[0]);
      const float u = sigmoidf_(v[1] + xx1 + ww[1]);
      const float n = tanhf(v[2] + xx2 + r * ww[2]);
      put(hnew, rb, c0, row, col, (1.f - u) * n + u * ww[3]);
      const size_t o = (size_t)row * R + col;
      rg[o] = r; ug[o] = u; ng[o] = n;
    }
  }
}

// (single product: the VRNN form, D2 == nullptr)
template <int NW, bool BF, int RT, class Red>
__device__ __forceinline__ void tile_dz_rt(const float* D, const float* WT, bool polled, const float* dz_add, int ld_add, bool add_polled, const DzIn& a,
                                           const Out& dqh, const Out& dph, int H, int Z, int r0, int c0, int B, Red& red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int col = c0 + (t & 15);
  int rowc[RT];
  float mq[RT], sq[RT], mp[RT], sp[RT], e[RT], rq[RT], rp[RT], c_raw[RT], c_fn[RT], e_add[RT], mqr[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = r0 + 16 * rt + (t >> 4);
    rowc[rt] = row < B ? row : r0;
    mq[rt] = 0.f; sq[rt] = 1.f; mp[rt] = 0.f; sp[rt] = 1.f; e[rt] = 0.f; rq[rt] = 0.f; rp[rt] = 0.f; c_raw[rt] = 0.f; c_fn[rt] = 0.f; e_add[rt] = 0.f; mqr[rt] = 0.f;
  }
  auto prefetch = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const size_t o = (size_t)rowc[rt] * Z + col;
      mq[rt] = a.mu_q[o]; sq[rt] = a.sd_q[o]; mp[rt] = a.mu_p[o]; sp[rt] = a.sd_p[o]; e[rt] = a.eps[o]; rq[rt] = a.raw_q[o]; rp[rt] = a.raw_p[o];
      if (a.residual == 2) mqr[rt] = a.muq_raw[o];
      if (a.c_fn != nullptr || a.c_raw != nullptr) {
        const bool live = (long long)a.t * a.stride < a.x_sl[rowc[rt]];
        c_raw[rt] = (live && a.c_raw != nullptr) ? a.c_raw[rowc[rt]] : 0.f;
        c_fn[rt] = (live && a.c_fn != nullptr) ? a.c_fn[rowc[rt]] : 0.f;
      }
      if (dz_add != nullptr && !add_polled) e_add[rt] = dz_add[(size_t)rowc[rt] * ld_add + col];
    }
  };
  f32x4 acc[RT][1];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.has_gemm) {  // uniform
    const float* const As[1] = {D};
    const float* const Ws[1] = {WT};
    const int la[1] = {H}, cs[1] = {c0};
    mgemm16_rt<NW, BF, RT, 1, 1, MapSame>(As, la, polled, r0, B, Ws, cs, H, acc, pl, prefetch);
  } else {
    prefetch();
  }
#pragma unroll
  for (int pr = 0; pr < RT / 2; ++pr) {
    if (r0 + 32 * pr >= B) break;
    float v[1] = {0.f};
    if (a.has_gemm) reduce_pair<1, NW>(acc[2 * pr], acc[2 * pr + 1], red(), v);
    const int hi = (int)(threadIdx.x >> 8), rt = 2 * pr + hi, rb = r0 + 16 * rt;
    const int row = rb + (t >> 4);
    const bool own = row < B;
#define PICK(x) (hi ? x[2 * pr + 1] : x[2 * pr])
    const int rc = PICK(rowc);
    const float mq_ = PICK(mq), sq_ = PICK(sq), mp_ = PICK(mp), sp_ = PICK(sp), e_ = PICK(e), rq_ = PICK(rq), rp_ = PICK(rp), craw = PICK(c_raw), cfn = PICK(c_fn),
                mqr_ = PICK(mqr);
    float add = PICK(e_add);
#undef PICK
    if (dz_add != nullptr && add_polled) {
      const rsrc_t rs[1] = {make_rsrc(dz_add)};
      const unsigned os[1] = {4u * ((unsigned)rc * (unsigned)ld_add + (unsigned)col)};
      float ws[1];
      poll_words<1>(rs, os, ws, own, pl);
      add = ws[0];
    }
    if (own) {
      const float dz = v[0] + add;
      const float d = mq_ - mp_, ip2 = 1.f / (sp_ * sp_);
      float coef = craw;
      if (cfn != 0.f) {
        const float k = logf(sp_) - logf(sq_) + (sq_ * sq_ + d * d) * 0.5f * ip2 - 0.5f;
        if (!(a.fn_floor > 0.f) || k > a.fn_floor) coef += cfn;
      }
      float g_muq = dz + coef * d * ip2;
      float g_sdq = dz * e_ + coef * (sq_ * ip2 - 1.f / sq_);
      float g_mup = -coef * d * ip2;
      float g_sdp = coef * (1.f / sp_ - (sq_ * sq_ + d * d) * ip2 / sp_);
      if (a.residual == 1) {
        g_mup += g_muq;
      } else if (a.residual == 2) {
        const float sqr = softplus_beta(rq_, a.beta, 1.f / a.beta) + a.sd_eps;
        const float pq = 1.f / (sqr * sqr), pp = ip2, var = sq_ * sq_;
        const float half_s3 = 0.5f * var * sq_;
        const float g_pq = g_muq * var * (mqr_ - mq_) - g_sdq * half_s3;
        const float g_pp = g_muq * var * (mp_ - mq_) - g_sdq * half_s3;
        g_mup += g_muq * var * pp;
        g_sdp += g_pp * (-2.f * pp / sp_);
        g_sdq = g_pq * (-2.f * pq / sqr);
        g_muq = g_muq * var * pq;
      }
      put(dqh, rb, c0, row, col, g_muq);
      put(dqh, rb, Z + c0, row, Z + col, g_sdq * sigmoidf_(a.beta * rq_));
      put(dph, rb, c0, row, col, g_mup);
      put(dph, rb, Z + c0, row, Z + col, g_sdp * sigmoidf_(a.beta * rp_));
    }
  }
}

template <int NW, bool BF, int RT, class Red>
__device__ __forceinline__ void tile_grub_rt(const GrubIn& a, int K, int R, int r0, int c0, int B, Red& red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int col = c0 + (t & 15);
  int rowc[RT];
  float r[RT], u[RT], n[RT], hn[RT], hp[RT], dd[RT], gadd[RT], g0[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = r0 + 16 * rt + (t >> 4);
    rowc[rt] = row < B ? row : r0;
    r[rt] = u[rt] = n[rt] = hn[rt] = hp[rt] = dd[rt] = gadd[rt] = g0[rt] = 0.f;
  }
  auto prefetch = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const size_t o = (size_t)rowc[rt] * R + col, o3 = (size_t)rowc[rt] * 3 * R + col;
      if (a.has_gin) g0[rt] = ld_sc1(make_rsrc(a.g_in), 4u * (unsigned)o);
      if (a.has_gates) {
        r[rt] = a.rg[o]; u[rt] = a.ug[o]; n[rt] = a.ng[o]; hn[rt] = a.gh[o3 + 2 * R];
        hp[rt] = a.hprev[(size_t)rowc[rt] * a.ldh + col];
        if (a.dd != nullptr) dd[rt] = a.dd[(size_t)rowc[rt] * a.ldh + col];
      }
      if (a.g_add != nullptr) gadd[rt] = a.g_add[(size_t)rowc[rt] * a.ld_gadd + col];
    }
  };
  f32x4 acc[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  if (!a.has_gemm) {
    prefetch();
  } else {
    const float* const As[2] = {a.D0, a.D1};
    const float* const Ws[2] = {a.W0, a.W1};
    const int la[2] = {0, 0}, cs[2] = {c0, c0};
    mgemm16_rt<NW, BF, RT, 2, 2, MapId>(As, la, true, r0, B, Ws, cs, K, acc, pl, prefetch);
  }
#pragma unroll
  for (int pr = 0; pr < RT / 2; ++pr) {
    if (r0 + 32 * pr >= B) break;
    float v[2] = {0.f, 0.f};
    if (a.has_gemm) reduce_pair<2, NW>(acc[2 * pr], acc[2 * pr + 1], red(), v);
    const int hi = (int)(threadIdx.x >> 8), rt = 2 * pr + hi, rb = r0 + 16 * rt;
    const int row = rb + (t >> 4);
    const bool own = row < B;
#define PICK(x) (hi ? x[2 * pr + 1] : x[2 * pr])
    const int rc = PICK(rowc);
    const float r_ = PICK(r), u_ = PICK(u), n_ = PICK(n), hn_ = PICK(hn), hp_ = PICK(hp), dd_ = PICK(dd), gadd_ = PICK(gadd);
    float g0_ = PICK(g0);
#undef PICK
    const size_t o = (size_t)rc * R + col;
    if (a.has_gin && __any(own && is_sentinel(g0_))) {
      const rsrc_t rs[1] = {make_rsrc(a.g_in)};
      const unsigned os[1] = {4u * (unsigned)o};
      float ws[1];
      poll_words<1>(rs, os, ws, own, pl);
      g0_ = ws[0];
    }
    if (own) {
      const float g = (g0_ + gadd_) + v[0] + v[1];
      if (!a.has_gates) {
        a.g_out[(size_t)row * R + col] = g;
      } else {
        const float dn_pre = g * (1.f - u_) * (1.f - n_ * n_);
        const float du_pre = g * (hp_ - n_) * u_ * (1.f - u_);
        const float dr_pre = dn_pre * hn_ * r_ * (1.f - r_);
        put(a.dgi, rb, c0, row, col, dr_pre); put(a.dgi, rb, R + c0, row, R + col, du_pre); put(a.dgi, rb, 2 * R + c0, row, 2 * R + col, dn_pre);
        put(a.dgh, rb, c0, row, col, dr_pre); put(a.dgh, rb, R + c0, row, R + col, du_pre); put(a.dgh, rb, 2 * R + c0, row, 2 * R + col, dn_pre * r_);
        st_sc1(make_rsrc(a.ga), 4u * ((unsigned)row * (unsigned)R + (unsigned)col), g * u_ + dd_);
      }
    }
  }
}

}  // namespace pchain
}  // namespace blvm
