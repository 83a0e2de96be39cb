// pchain.h — device primitives of the PERSISTENT recurrent chains (one launch per sequence instead of one per link).
//
// Why: a recurrent step of VRNN / SRNN / RSSM / GRU / LSTM is a chain of dependent [B,K]x[K,N] products with B = 8..64 rows.
// Launched link by link (stages.h) a link costs ~4 us on MI355X — 1.5 us of kernel boundary plus the argument fetch and an
// operand fetch from beyond the per-XCD L2, which every boundary invalidates — against 0.2-0.4 us of fp32 MFMA work.  Here the
// whole sequence is ONE launch of G co-resident workgroups (G <= number of CUs, one per CU).  Every link's 16x16 output tiles
// are dealt over the workgroups; a workgroup walks the (step, link) program in order and for each of its tiles
//   1. requests its weight fragments (T16 layout; weights never change inside the launch, so this runs ahead of the wait),
//   2. POLLS its activation operand: the producing workgroups store their output words write-through (`sc1`), the consumer
//      re-reads its own MFMA fragments with L1-bypassing `sc1` loads until no word holds the SENTINEL the host filled the
//      buffer with (0xFFFFFFFF, a NaN payload no arithmetic produces).  The data is its own flag: every 4-byte word is validated
//      by itself, so no ordering between words, no flag, no fence and no barrier is needed (MI355X_MICROARCH.md "handoff-1to1":
//      0.8-1.0 us for <= 4 KB, cross-XCD +0.1-0.3) and the successful poll IS the operand fetch,
//   3. MFMAs (K split over the NW waves), LDS reduction, fused epilogue, `sc1` stores.
// Every buffer a link writes is a per-step slab ([T', B, F]: the activations kept for BPTT anyway), so no location is written
// twice in a launch: no re-arming, no WAR hazard.  Progress: all G workgroups are resident and every workgroup processes its tiles
// in (step, link) order, so the earliest unfinished tile never waits on a later one.  Every spin is bounded: a wave that gives up
// raises the launch's abort word, every other spin sees it and the grid drains (the host reports BLVM_ELAUNCH).
#pragma once
#include "common.h"

namespace blvm {
namespace pchain {

constexpr unsigned SENTINEL = 0xFFFFFFFFu;          // hipMemsetAsync(buf, 0xFF, bytes)
constexpr unsigned SPIN_LIMIT = 1u << 22;           // polls before a wave gives up (~seconds)

// ---- write-through / L1-bypassing accesses ---------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 ld_sc1_x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// L1-bypassing but L2-served: only for words whose producer runs on THIS XCD (same L2) and stores them plainly
__device__ __forceinline__ f32x4 ld_nt_x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// the asm loads above are invisible to the compiler's own s_waitcnt insertion: wait explicitly, with the loaded registers as
// operands so that no use of them can be scheduled above the wait
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0(f32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void wait_vm0(float& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }

__device__ __forceinline__ bool is_sentinel(float x) { return __float_as_uint(x) == SENTINEL; }
__device__ __forceinline__ bool any_sentinel(const f32x4& v) {
  return is_sentinel(v[0]) | is_sentinel(v[1]) | is_sentinel(v[2]) | is_sentinel(v[3]);
}

// launch-wide state.  `dev[0]` (device memory, shared by all launches of the process, never reset): the epoch of the last launch in
// which a wave gave up; a launch is aborted when it reads its own epoch there.  `host` (pinned host memory mapped into the device,
// written only on failure): [0] number of aborted launches so far, [1] code of the spin that failed last (blvm_async_errors()).
struct Ctl {
  unsigned* dev;
  unsigned* host;
  unsigned epoch;
  __device__ __forceinline__ bool aborted() const { return __hip_atomic_load(dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch; }
  __device__ __forceinline__ void abort(unsigned code) const {
    if (__hip_atomic_exchange(dev, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __hip_atomic_store(host + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
};

// One bounded-spin bookkeeping step of a wave (uniform): returns true when the wave must stop waiting.
__device__ __forceinline__ bool spin_tick(unsigned& spins, const Ctl& ctl, unsigned code, bool& dead) {
  ++spins;
  if ((spins & 255u) == 0) {
    if (ctl.aborted()) { dead = true; return true; }
    if (spins >= SPIN_LIMIT) { ctl.abort(code); dead = true; return true; }
  }
  return false;
}

// per-wave waiting state of the launch: where to report, whether this wave has given up, how long to nap between polls (tiles off
// the critical path poll gently: every poll of a not-yet-written line is a fabric transaction that competes with the critical ones)
struct Poll {
  Ctl ctl;
  unsigned code;
  bool dead;
  int nap;
  bool local = false;  // operands come from workgroups on this XCD through its L2 (plain stores, nt loads) — probe only so far
#ifdef PCHAIN_TPROF  // variant build: wall-clock anatomy of tile_lin (wave 0): see tile_lin
  unsigned long long t_first = 0, t_ok = 0, tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned polls = 0;
#endif
  __device__ __forceinline__ void sleep() const {
    if (nap > 8) __builtin_amdgcn_s_sleep(32);
    else __builtin_amdgcn_s_sleep(1);
  }
};

// which activation operand product g multiplies
struct MapSame { static constexpr int of(int) { return 0; } };        // every product reads A[0]           (GRU / LSTM gates)
struct MapId { static constexpr int of(int g) { return g; } };        // product g reads A[g]               (dh: DP0 | DQ0)
struct MapPairs { static constexpr int of(int g) { return g >> 1; } };  // products 2a, 2a+1 read A[a]       (both Gaussian heads)

// ---- polled 16x16xK products ------------------------------------------------------------------------------------------------
// acc[g] += A[AMap(g)][r0+i][k] W[g][c0[g]+j][k] over the k-chunks owned by `wave` (chunk = 16 k, waves interleave chunks), W in
// the T16 operand layout with row length K.  When `polled`, the A operands are produced by other workgroups of this launch: their
// fragments are re-read with sc1 loads until no word is the sentinel.  One trip = CH chunks: all weight fragments first (they do
// not depend on the wait), then the activation fragments, then 4*CH*G MFMAs; a wave's chunk sum runs in ascending k.
template <int NW, int GA, int G, class AMap, int CH>
__device__ __forceinline__ void mgemm_trip(const float* const (&ap)[GA], const float* const (&wp)[G], int kc, bool aok, bool polled,
                                           f32x4 (&acc)[G], Poll& pl) {
  constexpr int STEP = NW * 16;
  f32x4 w[G][CH], a[GA][CH];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int u = 0; u < CH; ++u) w[g][u] = *reinterpret_cast<const f32x4*>(wp[g] + 16 * (size_t)(kc + u * STEP));
  if (!polled) {
#pragma unroll
    for (int g = 0; g < GA; ++g)
#pragma unroll
      for (int u = 0; u < CH; ++u) a[g][u] = *reinterpret_cast<const f32x4*>(ap[g] + kc + u * STEP);
  } else {
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int g = 0; g < GA; ++g)
#pragma unroll
        for (int u = 0; u < CH; ++u) a[g][u] = pl.local ? ld_nt_x4(ap[g] + kc + u * STEP) : ld_sc1_x4(ap[g] + kc + u * STEP);
#pragma unroll
      for (int g = 0; g < GA; ++g)
#pragma unroll
        for (int u = 0; u < CH; ++u) wait_vm0(a[g][u]);
#ifdef PCHAIN_TPROF
      if (spins == 0 && pl.t_first == 0) pl.t_first = wall_clock64();
      pl.polls++;
#endif
      bool bad = false;
#pragma unroll
      for (int g = 0; g < GA; ++g)
#pragma unroll
        for (int u = 0; u < CH; ++u) bad |= any_sentinel(a[g][u]);
      if (!__any(bad && aok) || pl.dead) break;
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
#ifdef PCHAIN_TPROF
    pl.t_ok = wall_clock64();
#endif
  }
  if (!aok) {
#pragma unroll
    for (int g = 0; g < GA; ++g)
#pragma unroll
      for (int u = 0; u < CH; ++u) a[g][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < CH; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[AMap::of(g)][u][e], w[g][u][e], acc[g], 0, 0, 0);
}

template <int NW, int GA, int G, class AMap>
__device__ __forceinline__ void mgemm16(const float* const (&A)[GA], const int (&lda)[GA], bool polled, int r0, int nrows,
                                        const float* const (&W)[G], const int (&c0)[G], int K, f32x4 (&acc)[G], Poll& pl) {
  constexpr int STEP = NW * 16;
  // fragment registers of a trip: 4 * CH * (G + GA); keep it <= 64
  constexpr int FR = (NW >= 16 ? 8 : 16) / (G + GA);  // 1024-thread workgroups have 128 VGPRs per lane
  constexpr int MAXCH = FR >= 8 ? 8 : (FR >= 4 ? 4 : (FR >= 2 ? 2 : 1));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane & 15, q = lane >> 4;
  const bool aok = (r0 + rr) < nrows;
  const float* ap[GA];
  const float* wp[G];
#pragma unroll
  for (int g = 0; g < GA; ++g) ap[g] = A[g] + (size_t)(aok ? r0 + rr : r0) * lda[g] + 4 * q;
#pragma unroll
  for (int g = 0; g < G; ++g) wp[g] = W[g] + (size_t)c0[g] * K + 4 * lane;
  int nch = (K / 16 - wave + NW - 1) / NW;  // chunks wave, wave + NW, ... below K / 16 (wave-uniform)
  int kc = wave * 16;
  while (nch > 0) {
    if (MAXCH >= 8 && nch >= 8) { mgemm_trip<NW, GA, G, AMap, (MAXCH >= 8 ? 8 : 1)>(ap, wp, kc, aok, polled, acc, pl); kc += 8 * STEP; nch -= 8; }
    else if (MAXCH >= 8 && nch >= 6) { mgemm_trip<NW, GA, G, AMap, (MAXCH >= 8 ? 6 : 1)>(ap, wp, kc, aok, polled, acc, pl); kc += 6 * STEP; nch -= 6; }
    else if (MAXCH >= 4 && nch >= 4) { mgemm_trip<NW, GA, G, AMap, (MAXCH >= 4 ? 4 : 1)>(ap, wp, kc, aok, polled, acc, pl); kc += 4 * STEP; nch -= 4; }
    else if (MAXCH >= 4 && nch >= 3) { mgemm_trip<NW, GA, G, AMap, (MAXCH >= 4 ? 3 : 1)>(ap, wp, kc, aok, polled, acc, pl); kc += 3 * STEP; nch -= 3; }
    else if (MAXCH >= 2 && nch >= 2) { mgemm_trip<NW, GA, G, AMap, (MAXCH >= 2 ? 2 : 1)>(ap, wp, kc, aok, polled, acc, pl); kc += 2 * STEP; nch -= 2; }
    else { mgemm_trip<NW, GA, G, AMap, 1>(ap, wp, kc, aok, polled, acc, pl); kc += STEP; nch -= 1; }
  }
}

// N epilogue words that other workgroups of this launch produce (sc1 loads until none is the sentinel); `need`: this lane uses them
template <int N>
__device__ __forceinline__ void poll_words(const float* const (&p)[N], float (&v)[N], bool need, Poll& pl) {
  unsigned spins = 0;
  for (;;) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ld_sc1(p[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) wait_vm0(v[i]);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < N; ++i) bad |= is_sentinel(v[i]);
    if (!__any(bad && need) || pl.dead) break;
    if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
    pl.sleep();
  }
}

// tiles [0, ntiles) of a segment are dealt over the workgroups [wg0, wg0 + nwg): tile i belongs to wg0 + i % nwg
__device__ __forceinline__ int first_tile(int w, int wg0, int nwg, int ntiles) {
  return (w >= wg0 && w < wg0 + nwg) ? w - wg0 : ntiles;
}

// The (row tile, column tile) pairs of a link for workgroup `w` of the range [wg0, wg0 + nwg).
//   xcd == false: tile i = c * rt + r belongs to wg0 + i % nwg.
//   xcd == true : workgroups are dealt round-robin over the 8 XCDs (observed; a wrong guess only costs speed), so workgroup wl of
//   the range sits on XCD wl % 8 with nwg / 8 peers: column tile c goes to XCD c % 8 with ALL its row tiles, so an XCD's L2 holds
//   1/8 of every weight matrix instead of all of it.  (nwg must be a multiple of 8.)
struct TileIter {
  int rt, ct, j, step, x, n;
  bool xcd;
  __host__ __device__ __forceinline__ TileIter(int w, int wg0, int nwg, int rt_, int ct_, bool xcd_) : rt(rt_), ct(ct_), xcd(xcd_) {
    const int wl = w - wg0;
    const bool in = w >= wg0 && wl < nwg;
    if (xcd) {
      x = wl & 7; step = nwg >> 3; j = wl >> 3;
      n = x < ct ? ((ct - x + 7) >> 3) * rt : 0;  // tiles of XCD x
    } else {
      x = 0; step = nwg; j = wl;
      n = rt * ct;
    }
    if (!in || step < 1) { j = 0; n = 0; step = 1; x = 0; }  // not a member of the range: no tiles
  }
  __host__ __device__ __forceinline__ bool valid() const { return j < n; }
  __host__ __device__ __forceinline__ void next() { j += step; }
  __host__ __device__ __forceinline__ int r0() const { return (j % rt) * 16; }
  __host__ __device__ __forceinline__ int c() const { return xcd ? x + 8 * (j / rt) : j / rt; }
};

// Cheap wait in front of a polled product: ONE wave polls one word of every 16-column producer tile of A[r0 .. r0+15][0 .. K) (a
// 4-byte sc1 load per producer and poll instead of the whole operand by every wave); the caller's barrier then releases the other
// waves into the validating operand poll, which normally succeeds at once.  Trades one memory round trip for far less poll traffic.
__device__ __forceinline__ void canary_wait(const float* A, int lda, int r0, int nrows, int K, Poll& pl) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x, np = K >> 4;
    const int rl = min(r0 + 15, nrows - 1);
    unsigned spins = 0;
    for (;;) {
      bool bad = false;
      for (int p0 = 0; p0 < np; p0 += 64) {
        const int pr = p0 + lane;
        float v = ld_sc1(A + (size_t)rl * lda + 16 * (pr < np ? pr : 0) + 15);
        wait_vm0(v);
        bad |= (pr < np) && is_sentinel(v);
      }
      if (!__any(bad) || pl.dead) break;
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
  }
  __syncthreads();
}

// ---- link tiles -------------------------------------------------------------------------------------------------------------
// Every tile function is called by ALL threads of the workgroup (NW * 64), contains exactly one workgroup barrier, and leaves its
// LDS scratch readable until the next-but-one tile (callers alternate between two scratch buffers).

// out = gate(act(A W^T + bias + add)):  bias [ncols] or null; add [B, ldadd] or null (add_polled: produced inside this launch);
// relu: act = leaky ReLU with `slope`; gate [B, ldgate] or null: result *= (gate > 0 ? 1 : slope) — the backward of that activation.
// out_sc1: the output is read by other workgroups of this launch.
template <int NW>
__device__ __forceinline__ void tile_lin(const float* A, int lda, bool a_polled, const float* W, int K, const float* bias,
                                         const float* add, int ldadd, bool add_polled, const float* gate, int ldgate, bool relu,
                                         float slope, float* out, int ldo, bool out_sc1, int r0, int c0, int B, float* red,
                                         Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;  // clamped row: the prefetches below are unconditional
#ifdef PCHAIN_TPROF
  const unsigned long long t0 = wall_clock64();
  pl.t_first = 0; pl.polls = 0;
#endif
  const float e_bias = bias ? bias[col] : 0.f;
  const float e_gate = gate ? gate[(size_t)rowc * ldgate + col] : 1.f;
  float e_add = (add && !add_polled) ? add[(size_t)rowc * ldadd + col] : 0.f;
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  {
    const float* const As[1] = {A};
    const float* const Ws[1] = {W};
    const int la[1] = {lda}, cs[1] = {c0};
    mgemm16<NW, 1, 1, MapSame>(As, la, a_polled, r0, B, Ws, cs, K, acc, pl);
  }
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
#ifdef PCHAIN_TPROF
  if (a_polled && pl.nap == 1) {  // critical tiles only: [0] tiles, [1] start -> first poll back, [2] -> poll ok, [3] -> reduced, [4] polls
    const unsigned long long t3 = wall_clock64();
    pl.tp[0] += 1; pl.tp[1] += pl.t_first - t0; pl.tp[2] += pl.t_ok - pl.t_first; pl.tp[3] += t3 - pl.t_ok; pl.tp[4] += pl.polls;
  }
#endif
  if (threadIdx.x >= 256) return;
  if (add && add_polled) {
    const float* const ps[1] = {add + (size_t)rowc * ldadd + col};
    float ws[1];
    poll_words<1>(ps, ws, own, pl);
    e_add = ws[0];
  }
  if (!own) return;
  float x = v[0] + e_bias + e_add;
  if (relu) x = x > 0.f ? x : x * slope;
  if (gate) x = e_gate > 0.f ? x : x * slope;
  if (out_sc1) st_sc1(out + (size_t)row * ldo + col, x);
  else out[(size_t)row * ldo + col] = x;
}

// Both Gaussian heads + posterior combination + reparameterised sample (stages.h head_stage_kernel): P, Q [B,H] are the last
// hidden layers of the prior / posterior MLP (polled), Wp, Wq [2Z,H] in T16.  z is read by other workgroups (sc1); the statistics
// are only read after the launch.  residual: 0 plain, 1 mu_q += mu_p, 2 precision-weighted, 3 generation (z ~ prior).
struct HeadOut {
  float *mu_p, *sd_p, *mu_q, *sd_q, *z, *raw_p, *raw_q, *muq_raw;  // [B,Z] slabs of this step; muq_raw may be null
};
template <int NW>
__device__ __forceinline__ void tile_head(const float* P, const float* Q, bool polled, const float* Wp, const float* bp, const float* Wq,
                                          const float* bq, const float* eps, const HeadOut& o, int H, int Z, int residual, float beta,
                                          float inv_beta, float sd_eps, int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const size_t oc = (size_t)(row < B ? row : r0) * Z + col;  // clamped: unconditional prefetch
  const float b0 = bp[col], b1 = bp[Z + col], b2 = bq[col], b3 = bq[Z + col];
  const float e = eps[oc];
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[2] = {P, Q};
    const float* const Ws[4] = {Wp, Wp, Wq, Wq};
    const int la[2] = {H, H}, cs[4] = {c0, Z + c0, c0, Z + c0};
    mgemm16<NW, 2, 4, MapPairs>(As, la, polled, r0, B, Ws, cs, H, acc, pl);
  }
  float v[4];
  reduce_tiles<4, NW>(acc, red, v);
  if (!own) return;
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
  st_sc1(o.z + oo, e * sqc + mq);  // randn_like(mu).mul(sd).add(mu)
  o.mu_p[oo] = mp; o.sd_p[oo] = sp; o.mu_q[oo] = mq; o.sd_q[oo] = sqc;
  o.raw_p[oo] = rp; o.raw_q[oo] = rq;
}

// GRU cell update of a [16 x 16] block of the state (vrnn.hip gru_stage_kernel): gi = X Wih^T (3 products, X [B,K] polled) + xg
// (state-independent part of the input projection incl. b_ih, computed before the launch) ; gh = h_prev Whh^T + b_hh was produced
// by another link of this launch (polled words), h_prev likewise.  Writes h_new (sc1) and the gates r, u, n (read after the launch).
template <int NW>
__device__ __forceinline__ void tile_gru(const float* X, int ldx, bool polled, const float* Wih, int K, const float* xg, const float* gh,
                                         const float* hprev, int ldh, int R, float* hnew, int ldn, float* rg, float* ug, float* ng,
                                         int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o3 = (size_t)rowc * 3 * R + col;
  const float x0 = xg ? xg[o3] : 0.f, x1 = xg ? xg[o3 + R] : 0.f, x2 = xg ? xg[o3 + 2 * R] : 0.f;
  // gh and h_prev were stored links ago: request them NOW (under the operand wait) and only re-poll in the rare case one is missing
  const float* const ps[4] = {gh + o3, gh + o3 + R, gh + o3 + 2 * R, hprev + (size_t)rowc * ldh + col};
  float w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = ld_sc1(ps[i]);
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[1] = {X};
    const float* const Ws[3] = {Wih, Wih, Wih};
    const int la[1] = {ldx}, cs[3] = {c0, R + c0, 2 * R + c0};
    mgemm16<NW, 1, 3, MapSame>(As, la, polled, r0, B, Ws, cs, K, acc, pl);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (threadIdx.x >= 256) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) wait_vm0(w[i]);
  if (__any(own && (is_sentinel(w[0]) | is_sentinel(w[1]) | is_sentinel(w[2]) | is_sentinel(w[3])))) poll_words<4>(ps, w, own, pl);
  if (!own) return;
  const float r = sigmoidf_(v[0] + x0 + w[0]);
  const float u = sigmoidf_(v[1] + x1 + w[1]);
  const float n = tanhf(v[2] + x2 + r * w[2]);
  st_sc1(hnew + (size_t)row * ldn + col, (1.f - u) * n + u * w[3]);
  const size_t o = (size_t)row * R + col;
  rg[o] = r; ug[o] = u; ng[o] = n;
}

// dz = D WT^T (+ D2 WT2^T) (+ dz_add), then back through rsample / posterior combination / KL (+ free nats) / softplus heads
// (stages.h dz_stage_kernel): D [B,H] (and D2) polled, WT [Z,H] in T16; writes the gradients wrt both heads' Linear outputs
// dqh, dph [B,2Z] (sc1: the next link multiplies them).
struct DzIn {
  const float *mu_q, *sd_q, *mu_p, *sd_p, *eps, *raw_q, *raw_p, *muq_raw;  // [B,Z] slabs of this step (saved by the forward)
  const int32_t* x_sl;
  const float *c_raw, *c_fn;  // [B] or null
  int t, stride, residual;
  float fn_floor, beta, sd_eps;
};
template <int NW>
__device__ __forceinline__ void tile_dz(const float* D, const float* WT, const float* D2, const float* WT2, bool polled, const float* dz_add,
                                        int ld_add, bool add_polled, const DzIn& a, float* dqh, float* dph, int H, int Z, int r0, int c0,
                                        int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * Z + col;
  const float mq = a.mu_q[o], sq = a.sd_q[o], mp = a.mu_p[o], sp = a.sd_p[o], e = a.eps[o], rq = a.raw_q[o], rp = a.raw_p[o];
  float c_raw = 0.f, c_fn = 0.f;
  if (a.c_fn != nullptr || a.c_raw != nullptr) {
    const bool live = (long long)a.t * a.stride < a.x_sl[rowc];
    c_raw = (live && a.c_raw != nullptr) ? a.c_raw[rowc] : 0.f;
    c_fn = (live && a.c_fn != nullptr) ? a.c_fn[rowc] : 0.f;
  }
  float e_add = (dz_add != nullptr && !add_polled) ? dz_add[(size_t)rowc * ld_add + col] : 0.f;
  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (D2 != nullptr) {
    const float* const As[2] = {D, D2};
    const float* const Ws[2] = {WT, WT2};
    const int la[2] = {H, H}, cs[2] = {c0, c0};
    mgemm16<NW, 2, 2, MapId>(As, la, polled, r0, B, Ws, cs, H, acc, pl);
  } else {
    f32x4 a1[1] = {acc[0]};
    const float* const As[1] = {D};
    const float* const Ws[1] = {WT};
    const int la[1] = {H}, cs[1] = {c0};
    mgemm16<NW, 1, 1, MapSame>(As, la, polled, r0, B, Ws, cs, H, a1, pl);
    acc[0] = a1[0];
  }
  float v[2];
  reduce_tiles<2, NW>(acc, red, v);
  if (threadIdx.x >= 256) return;
  if (dz_add != nullptr && add_polled) {
    const float* const ps[1] = {dz_add + (size_t)rowc * ld_add + col};
    float ws[1];
    poll_words<1>(ps, ws, own, pl);
    e_add = ws[0];
  }
  if (!own) return;
  const float dz = v[0] + v[1] + e_add;
  const float d = mq - mp, ip2 = 1.f / (sp * sp);
  float coef = c_raw;
  if (c_fn != 0.f) {
    const float k = logf(sp) - logf(sq) + (sq * sq + d * d) * 0.5f * ip2 - 0.5f;
    if (!(a.fn_floor > 0.f) || k > a.fn_floor) coef += c_fn;
  }
  float g_muq = dz + coef * d * ip2;
  float g_sdq = dz * e + coef * (sq * ip2 - 1.f / sq);
  float g_mup = -coef * d * ip2;
  float g_sdp = coef * (1.f / sp - (sq * sq + d * d) * ip2 / sp);
  if (a.residual == 1) {
    g_mup += g_muq;  // mu_q = mu_q' + mu_p
  } else if (a.residual == 2) {  // precision-weighted product of q' and p (stages.h)
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
  st_sc1(dqh + o2, g_muq);
  st_sc1(dqh + o2 + Z, g_sdq * sigmoidf_(a.beta * rq));
  st_sc1(dph + o2, g_mup);
  st_sc1(dph + o2 + Z, g_sdp * sigmoidf_(a.beta * rp));
}

}  // namespace pchain
}  // namespace blvm
