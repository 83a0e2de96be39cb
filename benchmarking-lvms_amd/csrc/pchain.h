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
// twice in a launch: no re-arming, no WAR hazard.  What the NEXT link multiplies is stored a second time in the T16 OPERAND LAYOUT
// (common.h: a 16-row x 16-column output tile IS one 1 KB block of it), so the consumer's polled fragment loads are contiguous
// 1 KB wave loads instead of 16 rows x 64 B (tools/cu_ingest.hip: 150 against 38 GB/s per workgroup); the row-major copy (plain
// stores) serves everything that reads after the launch.  Progress: all G workgroups are resident and every workgroup processes its tiles
// in (step, link) order, so the earliest unfinished tile never waits on a later one.  Every spin is bounded: a wave that gives up
// raises the launch's abort word, every other spin sees it and the grid drains (the host reports BLVM_ELAUNCH).
#pragma once
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace blvm {
namespace pchain {

constexpr unsigned SENTINEL = 0xFFFFFFFFu;          // hipMemsetAsync(buf, 0xFF, bytes)
constexpr unsigned SPIN_LIMIT = 1u << 22;           // polls before a wave gives up (~seconds)

// ---- write-through / L1-bypassing accesses ---------------------------------------------------------------------------------
// Buffer instructions with the sc1 cache bit through the compiler's builtins (aux = 16), NOT inline asm: the compiler then knows
// when a loaded value is available and places the waits itself.  With asm loads it believed the destination registers valid at
// issue: a copy it inserted between issue and wait (a loop-carried register, for one) read them early — results depended on timing.
// A buffer resource is (base pointer, 4 GB range); the lane's address is base + a 32-bit byte offset.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int AUX_SC1 = 16;
// (tools/pchain_probe.hip compiles the hand-off with other cache bits: -DPCHAIN_POLL_AUX=1 = sc0 loads, -DPCHAIN_STORE_AUX=0 = plain stores)
#ifndef PCHAIN_POLL_AUX
#define PCHAIN_POLL_AUX AUX_SC1
#endif
#ifndef PCHAIN_STORE_AUX
#define PCHAIN_STORE_AUX AUX_SC1
#endif
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {  // base must be wave-uniform
  // (said to the compiler, too: a pointer that reached here through a struct assigned under control flow counts as divergent, and
  // every buffer access through it became a waterfall loop over the "different" resources)
  const unsigned long long u = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1_x4(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, PCHAIN_POLL_AUX));
}
__device__ __forceinline__ float ld_sc1(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, PCHAIN_POLL_AUX));
}
__device__ __forceinline__ void st_sc1(rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, PCHAIN_STORE_AUX);
}

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
#ifdef PCHAIN_TPROF  // variant build: wall-clock anatomy of tile_lin (wave 0): see tile_lin
  unsigned long long t_first = 0, t_ok = 0, t_end = 0, tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned polls = 0;
#endif
  __device__ __forceinline__ void sleep() const {
    if (nap > 8) __builtin_amdgcn_s_sleep(32);
    else __builtin_amdgcn_s_sleep(1);
  }
};

// which activation operand product g multiplies
struct MapSame { static constexpr int of(int) { return 0; } static constexpr bool sum = false; };        // every product reads A[0]     (GRU / LSTM gates)
struct MapId { static constexpr int of(int g) { return g; } static constexpr bool sum = false; };        // product g reads A[g]         (dh: DP0 | DQ0)
struct MapPairs { static constexpr int of(int g) { return g >> 1; } static constexpr bool sum = false; };  // products 2a, 2a+1 read A[a] (both Gaussian heads)
// every product reads the SUM of all GA operands: the operand was produced in GA parts by links that split a long K among
// themselves (each part a full [rows, K] slab of partial sums) and is added up in the consumer's registers
struct MapSum { static constexpr int of(int) { return 0; } static constexpr bool sum = true; };

// ---- polled 16x16xK products ------------------------------------------------------------------------------------------------
// acc[g] += A[AMap(g)][r0+i][k] W[g][c0[g]+j][k] over the k-chunks owned by `wave` (chunk = 16 k, waves interleave chunks), W in
// the T16 operand layout with row length K.  When `polled`, the A operands are produced by other workgroups of this launch: A[g]
// is then the T16 copy ([rt*16, K] as 1 KB blocks) of the activation, `lda` is ignored, and the fragments are re-read with sc1
// loads until no word is the sentinel (rows >= nrows of the last row tile are never written and never looked at).  One trip = CH chunks: all weight fragments first (they do
// not depend on the wait), then the activation fragments, then 4*CH*G MFMAs; a wave's chunk sum runs in ascending k.
// `mid`: runs once per tile, right after the first trip's operand loads have been ISSUED and before anything waits on them: the
// place for the epilogue's own operand loads (bias, addend, gate, saved values), so that they travel with the operands instead of
// in front of them (a wait on them would also sit out the previous tile's write-through stores: gfx9 counts stores in vmcnt).
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};

// BF (the bf16-operand mode, blvm_set_operand_dtype): the weights were packed as bf16 (same T16 block order, 512 B per block) and the
// activation fragments are rounded to bf16 in registers — a lane's 4 floats of a chunk ARE the 4 k-values a lane feeds
// v_mfma_f32_16x16x16_bf16, so one MFMA replaces the four fp32 ones; accumulation, epilogues and everything stored stay fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
template <bool BF> struct WFrag { typedef f32x4 type; };
template <> struct WFrag<true> { typedef u32x2 type; };

template <int NW, bool BF, int GA, int G, class AMap, int CH, class Mid>
__device__ __forceinline__ void mgemm_trip(const rsrc_t (&ar)[GA], unsigned aoff, const float* const (&ap)[GA], const char* const (&wp)[G], int kc,
                                           bool aok, bool polled, f32x4 (&acc)[G], Poll& pl, Mid& mid, bool& mid_pending) {
  // Exactly CH chunks, no per-chunk guards: a guard around each chunk's load and the same guard around its sentinel check are one
  // region to the compiler, which then waits after EVERY chunk's load (one memory round trip per chunk instead of one per trip).
  // Polled operands are T16 slabs read through buffer resources (ar, byte offset aoff + 64 bytes per k), plain ones row-major (ap).
  constexpr int STEP = NW * 16;
  typedef typename WFrag<BF>::type wfrag;
  constexpr int ES = BF ? 2 : 4;  // bytes per weight element
  wfrag w[G][CH];
  f32x4 a[GA][CH];
#pragma unroll
  for (int u = 0; u < CH; ++u)
#pragma unroll
#ifdef PCHAIN_W_HOT  // timing experiment (garbage results): every weight fragment from the tile's first, cache-resident block —
                     // an upper bound on what weights kept next to the matrix pipe (LDS / registers) could save
    for (int g = 0; g < G; ++g) w[g][u] = *reinterpret_cast<const wfrag*>(wp[g] + (size_t)ES * 16 * (size_t)(u & 1));
#else
    for (int g = 0; g < G; ++g) w[g][u] = *reinterpret_cast<const wfrag*>(wp[g] + (size_t)ES * 16 * (size_t)(kc + u * STEP));
#endif
  if (!polled) {
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int g = 0; g < GA; ++g) a[g][u] = *reinterpret_cast<const f32x4*>(ap[g] + kc + u * STEP);
    if (mid_pending) { mid(); mid_pending = false; }
  } else {
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int g = 0; g < GA; ++g) a[g][u] = ld_sc1_x4(ar[g], aoff + 64u * (unsigned)(kc + u * STEP));
      if (mid_pending) { mid(); mid_pending = false; }  // behind the first poll's loads, in front of the first wait
      bool bad = false;
#pragma unroll
      for (int u = 0; u < CH; ++u)
#pragma unroll
        for (int g = 0; g < GA; ++g) bad |= any_sentinel(a[g][u]);
#ifdef PCHAIN_TPROF
      if (spins == 0 && pl.t_first == 0) pl.t_first = wall_clock64();
      pl.polls++;
#endif
      if (!__any(bad && aok) || pl.dead) break;
#ifdef PCHAIN_NOWAIT  // timing experiment: never wait (results are garbage): what the tiles cost without the hand-offs
      break;
#endif
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
#ifdef PCHAIN_TPROF
    pl.t_ok = wall_clock64();
#endif
  }
  if constexpr (AMap::sum) {
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int ga = 1; ga < GA; ++ga) a[0][u] += a[ga][u];
  }
  if constexpr (BF) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      s16x4 ab[GA];
#pragma unroll
      for (int ga = 0; ga < GA; ++ga) {
        const f32x4 x = a[ga][u];
        const u32x2 q = {aok ? pk_bf16(x[0], x[1]) : 0u, aok ? pk_bf16(x[2], x[3]) : 0u};
        ab[ga] = __builtin_bit_cast(s16x4, q);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[AMap::of(g)], __builtin_bit_cast(s16x4, w[g][u]), acc[g], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aok ? a[AMap::of(g)][u][e] : 0.f, w[g][u][e], acc[g], 0, 0, 0);
  }
}

template <int NW, bool BF, int GA, int G, class AMap, class Mid = NoMid>
__device__ __forceinline__ void mgemm16(const float* const (&A)[GA], const int (&lda)[GA], bool polled, int r0, int nrows,
                                        const float* const (&W)[G], const int (&c0)[G], int K, f32x4 (&acc)[G], Poll& pl, Mid mid = Mid(),
                                        int a_width = 0, int w_width = 0) {  // a_width / w_width: columns of the polled T16 slab / of the packed weight rows when wider than K (the product covers a K-range of them; the pointers start at the range)
  constexpr int STEP = NW * 16;
  // fragment registers of a trip: 4 * CH * (G + GA); trips of 6 / 4 / 2 / 1 chunks (K = 256, 512, 1536 on 8 waves: 2, 4, 6 + 6)
  // (16 waves share the K of a tile two ways finer and have half the registers each: trips of at most 3 -> 2 / 1 chunks)
  constexpr int FR = (NW >= 16 ? 6 : 12) / (G + GA);
  constexpr int MAXCH = FR >= 6 ? 6 : (FR >= 4 ? 4 : (FR >= 2 ? 2 : 1));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane & 15, q = lane >> 4;
  const bool aok = (r0 + rr) < nrows;
  rsrc_t ar[GA];
  const float* ap[GA];
  const char* wp[G];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    ar[g] = make_rsrc(A[g]);
    ap[g] = A[g] + (size_t)(aok ? r0 + rr : r0) * (polled ? 0 : lda[g]) + 4 * q;
  }
  const unsigned aoff = 4u * ((unsigned)(r0 >> 4) * 16u * (unsigned)(a_width > 0 ? a_width : K) + 4u * (unsigned)lane);  // T16: row tile's slab + this lane's fragment
#pragma unroll
  for (int g = 0; g < G; ++g) wp[g] = reinterpret_cast<const char*>(W[g]) + (BF ? 2 : 4) * ((size_t)c0[g] * (w_width > 0 ? w_width : K) + 4 * lane);
  int nch = (K / 16 - wave + NW - 1) / NW;  // chunks wave, wave + NW, ... below K / 16 (wave-uniform)
  int kc = wave * 16;
  bool mid_pending = true;
  if constexpr (MAXCH >= 6) for (; nch >= 6; nch -= 6, kc += 6 * STEP) mgemm_trip<NW, BF, GA, G, AMap, 6>(ar, aoff, ap, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if constexpr (MAXCH >= 4) for (; nch >= 4; nch -= 4, kc += 4 * STEP) mgemm_trip<NW, BF, GA, G, AMap, 4>(ar, aoff, ap, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if constexpr (MAXCH >= 2) for (; nch >= 2; nch -= 2, kc += 2 * STEP) mgemm_trip<NW, BF, GA, G, AMap, 2>(ar, aoff, ap, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  for (; nch >= 1; nch -= 1, kc += STEP) mgemm_trip<NW, BF, GA, G, AMap, 1>(ar, aoff, ap, wp, kc, aok, polled, acc, pl, mid, mid_pending);
  if (mid_pending) mid();  // a wave without chunks
}

// N epilogue words that other workgroups of this launch produce (sc1 loads until none is the sentinel); `need`: this lane uses them
template <int N>
__device__ __forceinline__ void poll_words(const rsrc_t (&r)[N], const unsigned (&off)[N], float (&v)[N], bool need, Poll& pl) {
  unsigned spins = 0;
  for (;;) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ld_sc1(r[i], off[i]);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < N; ++i) bad |= is_sentinel(v[i]);
    if (!__any(bad && need) || pl.dead) break;
#ifdef PCHAIN_NOWAIT
    break;
#endif
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

// Cheap wait in front of a polled product for tiles OFF the critical path: one wave polls ONE word of every 1 KB block of the
// T16 operand A16[r0 .. r0+15][0 .. K) instead of every wave polling its fragments; the barrier then releases the other waves into
// the validating operand poll, which normally succeeds at once.  Costs a round trip, saves the fabric most of the idle polling.
__device__ __forceinline__ void canary_wait(const float* A16, int r0, int K, Poll& pl, int a_width = 0) {  // a_width: as mgemm16
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x, np = K >> 4;
    const rsrc_t r = make_rsrc(A16 + (size_t)(r0 >> 4) * 16 * (a_width > 0 ? a_width : K));  // word 0 of a block = row r0, always written
    unsigned spins = 0;
    for (;;) {
      bool bad = false;
      for (int p0 = 0; p0 < np; p0 += 64) {
        const int pr = p0 + lane;
        const float v = ld_sc1(r, 1024u * (unsigned)(pr < np ? pr : 0));
        bad |= (pr < np) && is_sentinel(v);
      }
      if (!__any(bad) || pl.dead) break;
#ifdef PCHAIN_NOWAIT
      break;
#endif
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
  }
  __syncthreads();
}

// Where a link's [16 x 16] output tile goes: a row-major copy (plain stores; `rm_sc1`: other workgroups poll single words of it)
// and / or the T16 copy the next link multiplies (always sc1).  Either pointer may be null.
struct Out {
  float* rm;
  int ld;
  bool rm_sc1;
  float* x16;  // [rt*16, 16*n16] in 1 KB blocks
  int n16;
  float* x16b = nullptr;  // a second consumer's T16 operand (a concatenated input, e.g. cat[phi, h]: the pointer starts at this
  int n16b = 0;           // output's first block of a row tile, n16b = chunks per row of the whole concatenation)
};
__device__ __forceinline__ Out out_rm(float* p, int ld, bool sc1 = false) { return Out{p, ld, sc1, nullptr, 0}; }
__device__ __forceinline__ Out out_both(float* p, int ld, float* x16, int n16, bool sc1 = false) { return Out{p, ld, sc1, x16, n16}; }
__device__ __forceinline__ void put(const Out& o, int r0, int c0, int row, int col, float x) {
  if (o.rm != nullptr) {
    if (o.rm_sc1) st_sc1(make_rsrc(o.rm), 4u * ((unsigned)row * (unsigned)o.ld + (unsigned)col), x);
    else o.rm[(size_t)row * o.ld + col] = x;
  }
  const int rr = row - r0, cc = col - c0;
  const unsigned in_block = (unsigned)((rr + 16 * ((cc & 15) >> 2)) * 4 + (cc & 3));
  if (o.x16 != nullptr) st_sc1(make_rsrc(o.x16), 4u * ((((unsigned)(r0 >> 4) * (unsigned)o.n16 + (unsigned)((c0 + cc) >> 4)) << 8) + in_block), x);
  if (o.x16b != nullptr) st_sc1(make_rsrc(o.x16b), 4u * ((((unsigned)(r0 >> 4) * (unsigned)o.n16b + (unsigned)((c0 + cc) >> 4)) << 8) + in_block), x);
}

// ---- link tiles -------------------------------------------------------------------------------------------------------------
// Every tile function is called by ALL threads of the workgroup (NW * 64), contains exactly one workgroup barrier, and leaves its
// LDS scratch readable until the next-but-one tile (callers alternate between two scratch buffers).

// out = gate(act(A W^T + bias + add)):  bias [ncols] or null; add [B, ldadd] or null (add_polled: produced inside this launch);
// relu: act = leaky ReLU with `slope`; gate [B, ldgate] or null: result *= (gate > 0 ? 1 : slope) — the backward of that activation.
// What a linear tile needs only AFTER its operand loads are in flight (epilogue operands, output targets): produced by a functor
// the tile calls behind the first trip's loads, so that extracting them from a descriptor runs in the shadow of the operand wait.
struct LinLate {
  const float *bias, *add, *gate;
  int ldadd, ldgate;
  bool add_polled, relu;
  float slope;
  Out out;
};
template <int NW, bool BF, class Late>
__device__ __forceinline__ void tile_lin_late(const float* A, int lda, bool a_polled, const float* W, int K, Late& late, int r0, int c0, int B,
                                              float* red, Poll& pl, const float* A2 = nullptr, const float* A3 = nullptr, int w_width = 0) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;  // clamped row: the prefetches below are unconditional
#ifdef PCHAIN_TPROF
  const unsigned long long t0 = wall_clock64();
  pl.t_first = 0; pl.polls = 0;
#endif
  float e_bias = 0.f, e_gate = 1.f, e_add = 0.f;
  LinLate L;
  auto prefetch = [&]() {  // the epilogue's operands, requested behind the product's operand loads
    L = late();
    if (L.bias) e_bias = L.bias[col];
    if (L.gate) e_gate = L.gate[(size_t)rowc * L.ldgate + col];
    if (L.add && !L.add_polled) e_add = L.add[(size_t)rowc * L.ldadd + col];
  };
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  if (A2 != nullptr) {  // (uniform) the operand arrives as three partial-sum slabs
    const float* const As[3] = {A, A2, A3};
    const float* const Ws[1] = {W};
    const int la[3] = {0, 0, 0}, cs[1] = {c0};
    mgemm16<NW, BF, 3, 1, MapSum>(As, la, true, r0, B, Ws, cs, K, acc, pl, prefetch, lda, w_width);
  } else {
    const float* const As[1] = {A};
    const float* const Ws[1] = {W};
    const int la[1] = {lda}, cs[1] = {c0};
    mgemm16<NW, BF, 1, 1, MapSame>(As, la, a_polled, r0, B, Ws, cs, K, acc, pl, prefetch, a_polled ? lda : 0, w_width);  // polled: lda = slab width (0 = K)
  }
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
#ifdef PCHAIN_TPROF
  if (a_polled && pl.nap == 1) {  // critical tiles only: [0] tiles, [1] start -> first poll back, [2] -> poll ok, [3] -> reduced, [4] polls
    const unsigned long long t3 = wall_clock64();
    pl.tp[0] += 1; pl.tp[1] += pl.t_first - t0; pl.tp[2] += pl.t_ok - pl.t_first; pl.tp[3] += t3 - pl.t_ok; pl.tp[4] += pl.polls;
    if (pl.t_end != 0) pl.tp[6] += t0 - pl.t_end;  // end of the previous tile of any kind -> start of this one
  }
#endif
  if (threadIdx.x >= 256) return;
  if (L.add && L.add_polled) {
    const rsrc_t rs[1] = {make_rsrc(L.add)};
    const unsigned os[1] = {4u * ((unsigned)rowc * (unsigned)L.ldadd + (unsigned)col)};
    float ws[1];
    poll_words<1>(rs, os, ws, own, pl);
    e_add = ws[0];
  }
  if (!own) return;
  float x = v[0] + e_bias + e_add;
  if (L.relu) x = x > 0.f ? x : x * L.slope;
  if (L.gate) x = e_gate > 0.f ? x : x * L.slope;
  put(L.out, r0, c0, row, col, x);
#ifdef PCHAIN_TPROF
  if (a_polled && pl.nap == 1) { pl.t_end = wall_clock64(); }
#endif
}
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_lin(const float* A, int lda, bool a_polled, const float* W, int K, const float* bias,
                                         const float* add, int ldadd, bool add_polled, const float* gate, int ldgate, bool relu,
                                         float slope, const Out& out, int r0, int c0, int B, float* red, Poll& pl) {
  auto late = [&]() { return LinLate{bias, add, gate, ldadd, ldgate, add_polled, relu, slope, out}; };
  tile_lin_late<NW, BF>(A, lda, a_polled, W, K, late, r0, c0, B, red, pl);
}

// Both Gaussian heads + posterior combination + reparameterised sample (stages.h head_stage_kernel): P, Q [B,H] are the last
// hidden layers of the prior / posterior MLP (polled), Wp, Wq [2Z,H] in T16.  z is read by other workgroups (sc1); the statistics
// are only read after the launch.  residual: 0 plain, 1 mu_q += mu_p, 2 precision-weighted, 3 generation (z ~ prior).
struct HeadOut {
  float *mu_p, *sd_p, *mu_q, *sd_q, *raw_p, *raw_q, *muq_raw;  // [B,Z] slabs of this step; muq_raw may be null
  Out z;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_head(const float* P, const float* Q, bool polled, const float* Wp, const float* bp, const float* Wq,
                                          const float* bq, const float* eps, const HeadOut& o, int H, int Z, int residual, float beta,
                                          float inv_beta, float sd_eps, int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const size_t oc = (size_t)(row < B ? row : r0) * Z + col;  // clamped: unconditional prefetch
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, e = 0.f;
  auto prefetch = [&]() { b0 = bp[col]; b1 = bp[Z + col]; b2 = bq[col]; b3 = bq[Z + col]; e = eps[oc]; };
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[2] = {P, Q};
    const float* const Ws[4] = {Wp, Wp, Wq, Wq};
    const int la[2] = {H, H}, cs[4] = {c0, Z + c0, c0, Z + c0};
    mgemm16<NW, BF, 2, 4, MapPairs>(As, la, polled, r0, B, Ws, cs, H, acc, pl, prefetch);
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
  put(o.z, r0, c0, row, col, e * sqc + mq);  // randn_like(mu).mul(sd).add(mu)
  o.mu_p[oo] = mp; o.sd_p[oo] = sp; o.mu_q[oo] = mq; o.sd_q[oo] = sqc;
  o.raw_p[oo] = rp; o.raw_q[oo] = rq;
}

// GRU cell update of a [16 x 16] block of the state (vrnn.hip gru_stage_kernel, rssm.hip gru_cell_stage_kernel): gi = X Wih^T
// (3 products, X [B,K] polled) + xg (state-independent part of the input projection incl. b_ih, computed before the launch) and /
// or + b_ih ; gh = h_prev Whh^T + b_hh was produced by another link of this launch (polled words), h_prev likewise.  Writes h_new (sc1) and the gates r, u, n (read after the launch).
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_gru(const float* X, int ldx, bool polled, const float* Wih, int K, const float* xg, const float* bih,
                                         const float* gh, const float* hprev, int ldh, int R, const Out& hnew, float* rg, float* ug, float* ng,
                                         int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o3 = (size_t)rowc * 3 * R + col;
  // the state-independent part of the input projection: per row (xg, incl. b_ih: VRNN) and / or the bias alone (b_ih: RSSM);
  // gh and h_prev were stored links ago: requested under the operand wait, re-polled only in the rare case one is missing
  const rsrc_t rgh = make_rsrc(gh), rhp = make_rsrc(hprev);
  const rsrc_t ps[4] = {rgh, rgh, rgh, rhp};
  const unsigned po[4] = {4u * (unsigned)o3, 4u * (unsigned)(o3 + R), 4u * (unsigned)(o3 + 2 * R), 4u * ((unsigned)rowc * (unsigned)ldh + (unsigned)col)};
  float w[4] = {0.f, 0.f, 0.f, 0.f}, x0 = 0.f, x1 = 0.f, x2 = 0.f;
  auto prefetch = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = ld_sc1(ps[i], po[i]);
    x0 = (xg ? xg[o3] : 0.f) + (bih ? bih[col] : 0.f);
    x1 = (xg ? xg[o3 + R] : 0.f) + (bih ? bih[R + col] : 0.f);
    x2 = (xg ? xg[o3 + 2 * R] : 0.f) + (bih ? bih[2 * R + col] : 0.f);
  };
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[1] = {X};
    const float* const Ws[3] = {Wih, Wih, Wih};
    const int la[1] = {ldx}, cs[3] = {c0, R + c0, 2 * R + c0};
    mgemm16<NW, BF, 1, 3, MapSame>(As, la, polled, r0, B, Ws, cs, K, acc, pl, prefetch);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (threadIdx.x >= 256) return;
  if (__any(own && (is_sentinel(w[0]) | is_sentinel(w[1]) | is_sentinel(w[2]) | is_sentinel(w[3])))) poll_words<4>(ps, po, w, own, pl);
  if (!own) return;
  const float r = sigmoidf_(v[0] + x0 + w[0]);
  const float u = sigmoidf_(v[1] + x1 + w[1]);
  const float n = tanhf(v[2] + x2 + r * w[2]);
  put(hnew, r0, c0, row, col, (1.f - u) * n + u * w[3]);
  const size_t o = (size_t)row * R + col;
  rg[o] = r; ug[o] = u; ng[o] = n;
}

// dz = D WT^T (+ D2 WT2^T) (+ dz_add), then back through rsample / posterior combination / KL (+ free nats) / softplus heads
// (stages.h dz_stage_kernel): D [B,H] (and D2) polled, WT [Z,H] in T16; writes the gradients wrt both heads' Linear outputs
// dqh, dph [B,2Z] (row-major for the weight gradients, T16 for the next link).
struct DzIn {
  const float *mu_q, *sd_q, *mu_p, *sd_p, *eps, *raw_q, *raw_p, *muq_raw;  // [B,Z] slabs of this step (saved by the forward)
  const int32_t* x_sl;
  const float *c_raw, *c_fn;  // [B] or null
  int t, stride, residual;
  float fn_floor, beta, sd_eps;
  bool has_gemm = true;  // false: dz = dz_add alone (the last step of a chain whose z only feeds the next step)
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_dz(const float* D, const float* WT, const float* D2, const float* WT2, bool polled, const float* dz_add,
                                        int ld_add, bool add_polled, const DzIn& a, const Out& dqh, const Out& dph, int H, int Z, int r0, int c0,
                                        int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * Z + col;
  float mq = 0.f, sq = 1.f, mp = 0.f, sp = 1.f, e = 0.f, rq = 0.f, rp = 0.f, c_raw = 0.f, c_fn = 0.f, e_add = 0.f;
  auto prefetch = [&]() {  // everything the forward saved for this step, requested behind the product's operand loads
    mq = a.mu_q[o]; sq = a.sd_q[o]; mp = a.mu_p[o]; sp = a.sd_p[o]; e = a.eps[o]; rq = a.raw_q[o]; rp = a.raw_p[o];
    if (a.c_fn != nullptr || a.c_raw != nullptr) {
      const bool live = (long long)a.t * a.stride < a.x_sl[rowc];
      c_raw = (live && a.c_raw != nullptr) ? a.c_raw[rowc] : 0.f;
      c_fn = (live && a.c_fn != nullptr) ? a.c_fn[rowc] : 0.f;
    }
    if (dz_add != nullptr && !add_polled) e_add = dz_add[(size_t)rowc * ld_add + col];
  };
  float v[2] = {0.f, 0.f};
  if (a.has_gemm) {  // uniform
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (D2 != nullptr) {
      const float* const As[2] = {D, D2};
      const float* const Ws[2] = {WT, WT2};
      const int la[2] = {H, H}, cs[2] = {c0, c0};
      mgemm16<NW, BF, 2, 2, MapId>(As, la, polled, r0, B, Ws, cs, H, acc, pl, prefetch);
    } else {
      f32x4 a1[1] = {acc[0]};
      const float* const As[1] = {D};
      const float* const Ws[1] = {WT};
      const int la[1] = {H}, cs[1] = {c0};
      mgemm16<NW, BF, 1, 1, MapSame>(As, la, polled, r0, B, Ws, cs, H, a1, pl, prefetch);
      acc[0] = a1[0];
    }
    reduce_tiles<2, NW>(acc, red, v);
  } else {
    prefetch();
  }
  if (threadIdx.x >= 256) return;
  if (dz_add != nullptr && add_polled) {
    const rsrc_t rs[1] = {make_rsrc(dz_add)};
    const unsigned os[1] = {4u * ((unsigned)rowc * (unsigned)ld_add + (unsigned)col)};
    float ws[1];
    poll_words<1>(rs, os, ws, own, pl);
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
  // [B,2Z]: the mean half at column col, the scale half at Z + col (its T16 blocks follow the mean half's)
  put(dqh, r0, c0, row, col, g_muq);
  put(dqh, r0, Z + c0, row, Z + col, g_sdq * sigmoidf_(a.beta * rq));
  put(dph, r0, c0, row, col, g_mup);
  put(dph, r0, Z + c0, row, Z + col, g_sdp * sigmoidf_(a.beta * rp));
}

// Backward of a GRU state update fused with the products that complete the state gradient (vrnn.hip "B10"):
//   g = g_in + g_add + D0 W0^T + D1 W1^T  (has_gemm: the products; D0, D1 [B,K] polled, W0, W1 [R,K] T16; has_gin: g_in, one polled
//                                           word; g_add [B, ld_gadd]: a gradient reaching this state from outside the chain, or null)
//   gate derivatives of the step whose OUTPUT state g refers to (has_gates): r, u, n, hn = (h_prev W_hh^T + b_hh)_n, h_prev saved
//   by the forward; writes dgi = [dr, du, dn], dgh = [dr, du, dn * r] ([B,3R], row-major + T16) and ga = g * u + dd (polled words;
//   dd [B, ldh] = gradient reaching h_prev from outside the chain).  Without gates (the step before the first) g goes to g_out.
struct GrubIn {
  const float *D0, *D1, *W0, *W1, *g_in, *g_add;
  const float *rg, *ug, *ng, *gh, *hprev, *dd;  // dd may be null
  int ldh, ld_gadd;
  Out dgi, dgh;
  float *ga, *g_out;
  bool has_gemm, has_gin, has_gates;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_grub(const GrubIn& a, int K, int R, int r0, int c0, int B, float* red, Poll& pl) {
  const int tt = threadIdx.x & 255;
  const int row = r0 + (tt >> 4), col = c0 + (tt & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col, o3 = (size_t)rowc * 3 * R + col;
  float r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f, dd = 0.f, gadd = 0.f, g0 = 0.f;
  auto prefetch = [&]() {  // g_in was stored a step ago: requested under the operand wait like the saved values
    if (a.has_gin) g0 = ld_sc1(make_rsrc(a.g_in), 4u * (unsigned)o);
    if (a.has_gates) {
      r = a.rg[o]; u = a.ug[o]; n = a.ng[o]; hn = a.gh[o3 + 2 * R];
      hp = a.hprev[(size_t)rowc * a.ldh + col];
      if (a.dd != nullptr) dd = a.dd[(size_t)rowc * a.ldh + col];
    }
    if (a.g_add != nullptr) gadd = a.g_add[(size_t)rowc * a.ld_gadd + col];
  };
  float v[2] = {0.f, 0.f};
  if (!a.has_gemm) prefetch();
  if (a.has_gemm) {
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* const As[2] = {a.D0, a.D1};
    const float* const Ws[2] = {a.W0, a.W1};
    const int la[2] = {0, 0}, cs[2] = {c0, c0};
    mgemm16<NW, BF, 2, 2, MapId>(As, la, true, r0, B, Ws, cs, K, acc, pl, prefetch);
    reduce_tiles<2, NW>(acc, red, v);
  }
  if (threadIdx.x >= 256) return;
  if (a.has_gin && __any(own && is_sentinel(g0))) {
    const rsrc_t rs[1] = {make_rsrc(a.g_in)};
    const unsigned os[1] = {4u * (unsigned)o};
    float ws[1];
    poll_words<1>(rs, os, ws, own, pl);
    g0 = ws[0];
  }
  if (!own) return;
  const float g = (g0 + gadd) + v[0] + v[1];
  if (!a.has_gates) { a.g_out[(size_t)row * R + col] = g; return; }
  const float dn_pre = g * (1.f - u) * (1.f - n * n);
  const float du_pre = g * (hp - n) * u * (1.f - u);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  put(a.dgi, r0, c0, row, col, dr_pre); put(a.dgi, r0, R + c0, row, R + col, du_pre); put(a.dgi, r0, 2 * R + c0, row, 2 * R + col, dn_pre);
  put(a.dgh, r0, c0, row, col, dr_pre); put(a.dgh, r0, R + c0, row, R + col, du_pre); put(a.dgh, r0, 2 * R + c0, row, 2 * R + col, dn_pre * r);
  st_sc1(make_rsrc(a.ga), 4u * ((unsigned)row * (unsigned)R + (unsigned)col), g * u + dd);
}

// ---- whole GRU / LSTM sequences (rnn.hip: nn.GRU forward / per-row time-reversed, packed nn.LSTM) as one link per step ------------
// time index processed by `row` at recurrence step j: forward j; reversed: len-1-j inside the row's length, j in the right padding
__device__ __forceinline__ int seq_time_index(int j, int reverse, const int32_t* lens, int row) {
  if (!reverse) return j;
  const int n = lens[row];
  return j < n ? n - 1 - j : j;
}
struct GruSeqIn {
  const float *H16, *Whh, *bhh;  // state entering the step (T16 slab, polled), [3R,R] rows [r|z|n] in T16, [3R]
  const float* xg;               // [T,B,3R] input projection incl. b_ih, TIME indexed
  const int32_t* lens;           // [B] (reverse map) or null
  const float* hprev;            // [B,R] row-major: written by THIS thread one step earlier (plain)
  float* out;                    // element (idx,row,col) at out + idx*out_ts + row*out_ld + col
  float *rg, *ug, *ng, *ghn;     // [B,R] saves of this step
  long out_ts;
  int out_ld, j, reverse;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_gru_seq(const GruSeqIn& a, const Out& hnext, int R, int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col;
  int idx = 0;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, hp = 0.f;
  auto prefetch = [&]() {
    idx = seq_time_index(a.j, a.reverse, a.lens, rowc);
    const size_t ox = ((size_t)idx * B + rowc) * 3 * R + col;
    x0 = a.xg[ox]; x1 = a.xg[ox + R]; x2 = a.xg[ox + 2 * R];
    b0 = a.bhh[col]; b1 = a.bhh[R + col]; b2 = a.bhh[2 * R + col];
    hp = a.hprev[o];
  };
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[1] = {a.H16};
    const float* const Ws[3] = {a.Whh, a.Whh, a.Whh};
    const int la[1] = {0}, cs[3] = {c0, R + c0, 2 * R + c0};
    mgemm16<NW, BF, 1, 3, MapSame>(As, la, true, r0, B, Ws, cs, R, acc, pl, prefetch);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (!own) return;
  const float hn = v[2] + b2;
  const float r = sigmoidf_(x0 + v[0] + b0);
  const float u = sigmoidf_(x1 + v[1] + b1);
  const float n = tanhf(x2 + r * hn);
  const float h2 = (1.f - u) * n + u * hp;
  put(hnext, r0, c0, row, col, h2);
  a.out[(size_t)idx * a.out_ts + (size_t)row * a.out_ld + col] = h2;
  a.rg[o] = r; a.ug[o] = u; a.ng[o] = n; a.ghn[o] = hn;
}

struct GruSeqBwdIn {
  const float *DGHn16, *WhhT;           // hidden-projection grads of recurrence step j+1 (T16 slab, polled; K = 3R), [R,3R] in T16
  const float* dout;                    // time-indexed grad wrt the outputs (GruSeqIn::out addressing)
  const float *rg, *ug, *ng, *ghn, *hprev;  // saves of step j
  const int32_t* lens;
  float* G;                             // [B,R] running grad through the u-gate path: read and written by THIS thread only
  float* DGI;                           // [T,B,3R] TIME indexed
  float* dh0;                           // [B,R]: written when has_gates == 0
  long out_ts;
  int out_ld, j, reverse;
  bool has_gemm, has_gates;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_gru_seq_bwd(const GruSeqBwdIn& a, const Out& dgh, int R, int r0, int c0, int B, float* red, Poll& pl) {
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col;
  float g = 0.f, dout = 0.f, r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f;
  int idx = 0;
  auto prefetch = [&]() {
    g = a.G[o];
    if (a.has_gates) {  // uniform
      idx = seq_time_index(a.j, a.reverse, a.lens, rowc);
      dout = a.dout[(size_t)idx * a.out_ts + (size_t)rowc * a.out_ld + col];
      r = a.rg[o]; u = a.ug[o]; n = a.ng[o]; hn = a.ghn[o]; hp = a.hprev[o];
    }
  };
  float v[1] = {0.f};
  if (a.has_gemm) {  // uniform
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const float* const As[1] = {a.DGHn16};
    const float* const Ws[1] = {a.WhhT};
    const int la[1] = {0}, cs[1] = {c0};
    mgemm16<NW, BF, 1, 1, MapSame>(As, la, true, r0, B, Ws, cs, 3 * R, acc, pl, prefetch);
    reduce_tiles<1, NW>(acc, red, v);
  } else {
    prefetch();
    __syncthreads();  // (every tile has exactly one workgroup barrier)
  }
  if (!own) return;
  g += dout + v[0];
  if (!a.has_gates) { a.dh0[o] = g; return; }
  const float dn_pre = g * (1.f - u) * (1.f - n * n);
  const float du_pre = g * (hp - n) * u * (1.f - u);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  const size_t oi = ((size_t)idx * B + row) * 3 * R + col;
  a.DGI[oi] = dr_pre; a.DGI[oi + R] = du_pre; a.DGI[oi + 2 * R] = dn_pre;
  put(dgh, r0, c0, row, col, dr_pre);
  put(dgh, r0, R + c0, row, R + col, du_pre);
  put(dgh, r0, 2 * R + c0, row, 2 * R + col, dn_pre * r);
  a.G[o] = g * u;
}

struct LstmSeqIn {
  const float *H16, *Whh, *bhh;  // state entering the step (T16 slab, polled), [4H,H] rows [i|f|g|o] in T16, [4H]
  const float* xg;               // [B,4H] input projection of this step incl. b_ih
  const int32_t* lens;           // [B] valid steps per row (packed-sequence semantics) or null
  const float *hprev, *cprev;    // [B,H] row-major: written by THIS thread one step earlier (plain)
  float *cnext, *out, *gates;    // [B,H], [B,H] (zero past the row's length), [B,4H] saved i,f,g,o (zero where masked)
  int t;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_lstm_seq(const LstmSeqIn& a, const Out& hnext, int H, int r0, int c0, int B, float* red, Poll& pl) {
  const int tt = threadIdx.x & 255;
  const int row = r0 + (tt >> 4), col = c0 + (tt & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f, hp = 0.f, cp = 0.f;
  bool live = true;
  auto prefetch = [&]() {
    x0 = a.xg[o4] + a.bhh[col]; x1 = a.xg[o4 + H] + a.bhh[H + col];
    x2 = a.xg[o4 + 2 * H] + a.bhh[2 * H + col]; x3 = a.xg[o4 + 3 * H] + a.bhh[3 * H + col];
    hp = a.hprev[o]; cp = a.cprev[o];
    live = a.lens == nullptr || a.t < a.lens[rowc];
  };
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[1] = {a.H16};
    const float* const Ws[4] = {a.Whh, a.Whh, a.Whh, a.Whh};
    const int la[1] = {0}, cs[4] = {c0, H + c0, 2 * H + c0, 3 * H + c0};
    mgemm16<NW, BF, 1, 4, MapSame>(As, la, true, r0, B, Ws, cs, H, acc, pl, prefetch);
  }
  float v[4];
  reduce_tiles<4, NW>(acc, red, v);
  if (!own) return;
  const float i = sigmoidf_(v[0] + x0), f = sigmoidf_(v[1] + x1), g = tanhf(v[2] + x2), og = sigmoidf_(v[3] + x3);
  const float c2 = f * cp + i * g;
  const float h2 = og * tanhf(c2);
  a.cnext[o] = live ? c2 : cp;
  put(hnext, r0, c0, row, col, live ? h2 : hp);
  a.out[o] = live ? h2 : 0.f;
  a.gates[o4] = live ? i : 0.f;
  a.gates[o4 + H] = live ? f : 0.f;
  a.gates[o4 + 2 * H] = live ? g : 0.f;
  a.gates[o4 + 3 * H] = live ? og : 0.f;
}

struct LstmSeqBwdIn {
  const float *DGn16, *WhhT;  // gate pre-activation grads of step s+1 (T16 slab, polled; K = 4H), [H,4H] in T16
  const float *dout, *gates, *c_s, *c_s1;  // [B,H] grad wrt out_s, [B,4H] saved gates, cell state entering / leaving step s
  float* DC;                  // [B,H] running grad wrt the cell state: read and written by THIS thread only
  float* dh0;                 // [B,H]: written when has_gates == 0
  bool has_gemm, has_gates;
};
template <int NW, bool BF = false>
__device__ __forceinline__ void tile_lstm_seq_bwd(const LstmSeqBwdIn& a, const Out& dg, int H, int r0, int c0, int B, float* red, Poll& pl) {
  const int tt = threadIdx.x & 255;
  const int row = r0 + (tt >> 4), col = c0 + (tt & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col;
  float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cs = 0.f, cs1 = 0.f, dc = 0.f;
  auto prefetch = [&]() {
    if (a.has_gates) {  // uniform
      dh = a.dout[o];
      ig = a.gates[o4]; fg = a.gates[o4 + H]; gg = a.gates[o4 + 2 * H]; og = a.gates[o4 + 3 * H];
      cs = a.c_s[o]; cs1 = a.c_s1[o]; dc = a.DC[o];
    }
  };
  float v[1] = {0.f};
  if (a.has_gemm) {  // uniform
    f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const float* const As[1] = {a.DGn16};
    const float* const Ws[1] = {a.WhhT};
    const int la[1] = {0}, cs_[1] = {c0};
    mgemm16<NW, BF, 1, 1, MapSame>(As, la, true, r0, B, Ws, cs_, 4 * H, acc, pl, prefetch);
    reduce_tiles<1, NW>(acc, red, v);
  } else {
    prefetch();
    __syncthreads();
  }
  if (!own) return;
  dh += v[0];
  if (!a.has_gates) { a.dh0[o] = dh; return; }
  const float tc = tanhf(cs1);
  const float d_o = dh * tc;
  const float dct = dc + dh * og * (1.f - tc * tc);
  put(dg, r0, c0, row, col, dct * gg * ig * (1.f - ig));
  put(dg, r0, H + c0, row, H + col, dct * cs * fg * (1.f - fg));
  put(dg, r0, 2 * H + c0, row, 2 * H + col, dct * ig * (1.f - gg * gg));
  put(dg, r0, 3 * H + c0, row, 3 * H + col, d_o * og * (1.f - og));
  a.DC[o] = dct * fg;
}

// Sampling from a DMoL head during generation (`VRNN.generate`, blvm/models/vrnn.py:371-434: likelihood(dec) -> sample): a tile =
// 16 utterances x 4 samples of one frame stack.  dec [B, S*F] (F = 3 * num_mix = 30 head inputs per sample, row-major, polled words:
// the last decoder layer of this step) -> per sample the head's Linear(F -> F) -> Gumbel-max component pick with u, clamped
// logistic draw with v (dmol.hip mix_sample_kernel; both null: the mode) -> x [B, ldx] (plain) and the T16 copy the next step's
// encoder multiplies.  Wave w computes head outputs 4w .. 4w+3 of all 64 (utterance, sample) pairs (weights wave-uniform); wave 0
// then draws.  `lds`: >= 16*4*F + 64*32 floats.
template <int NW>
__device__ __forceinline__ void tile_dmol_sample(const float* dec, int ldd, const float* Wl, const float* bl, const float* u, const float* v, int S, int F,
                                                 int num_mix, float log_eps, const Out& xo, int r0, int s0, int B, float* lds, Poll& pl) {
  static_assert(NW == 8 || NW == 16, "waves");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* in = lds;               // [16 rows][4 * F]
  float* outp = lds + 16 * 4 * F;  // [64 pairs][32]
  const int rowlen = 4 * F;      // floats of a tile row (a multiple of 4: 16-byte pieces)
  const int pieces = 16 * rowlen / 4;
  const rsrc_t rd = make_rsrc(dec);
  {  // polled cooperative load of the tile's head inputs
    unsigned spins = 0;
    for (;;) {
      bool bad = false;
      for (int q = tid; q < pieces; q += NW * 64) {
        const int rr = q / (rowlen / 4), cq = q % (rowlen / 4);
        const bool ok = r0 + rr < B;
        const f32x4 x = ld_sc1_x4(rd, 4u * ((unsigned)(ok ? r0 + rr : r0) * (unsigned)ldd + (unsigned)(s0 * F + 4 * cq)));
        bad |= ok && any_sentinel(x);
        *reinterpret_cast<f32x4*>(in + rr * rowlen + 4 * cq) = x;
      }
      if (!__any(bad) || pl.dead) break;
#ifdef PCHAIN_NOWAIT
      break;
#endif
      if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
      pl.sleep();
    }
  }
  __syncthreads();  // (a wave that saw everything may pass while another still polls its pieces: the barrier orders them)
  {
    const int rr = lane >> 2, ss = lane & 3;
    const float* a = in + rr * rowlen + ss * F;
    if (wave * 4 < F) {
      float acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int jj = wave * 4 + j;
        acc[j] = jj < F ? bl[jj] : 0.f;
      }
      for (int k = 0; k < F; ++k) {
        const float ak = a[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int jj = wave * 4 + j;
          if (jj < F) acc[j] = fmaf(Wl[jj * F + k], ak, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (wave * 4 + j < F) outp[lane * 32 + wave * 4 + j] = acc[j];
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int rr = lane >> 2, ss = lane & 3, row = r0 + rr, smp = s0 + ss;
    if (row < B) {
      const float* p = outp + lane * 32;
      const size_t f = (size_t)row * S + smp;
      int best = 0;
      float bv = -INFINITY;
      for (int m = 0; m < num_mix; ++m) {
        float sc = p[m];
        if (u != nullptr) sc -= logf(-logf(u[f * num_mix + m]));
        if (sc > bv) { bv = sc; best = m; }  // first maximum, as torch.argmax
      }
      const float loc = p[num_mix + best], raw = p[2 * num_mix + best];
      float x = loc;
      if (v != nullptr) {
        const float vv = v[f];
        x = loc + expf(fmaxf(raw, log_eps)) * (logf(vv) - logf(1.f - vv));
        x = fminf(fmaxf(x, -1.f), 1.f);
      }
      put(xo, r0, s0 & ~15, row, smp, x);
    }
  }
  __syncthreads();  // the scratch is reused by the next tile
}

// =================================================================================================================================
// The program a persistent launch executes (pchain.hip: ONE kernel, one copy of every tile kind — the step program of a model is
// DATA, not code: with each link inlined at its own call site the VRNN kernel was 16 000 instructions, every link of a step ran
// from a cold instruction cache and cost ~1 us more than the same tile in a small kernel, tools/pchain_probe.hip).
// A launch walks steps s = 0 .. S-1; in every step each workgroup goes through the descriptors in order and runs the tiles that
// are its own (TileIter over [wg0, wg0 + nwg)).  A pointer of a descriptor is `p[k] + s * stride[sidx[k]]` (stride table of the
// program, entry 0 = 0: constants and null pointers); a backward sequence passes its last step's slabs and negative strides.
// =================================================================================================================================
enum Kind : int { K_LIN = 0, K_HEAD = 1, K_GRU = 2, K_DZ = 3, K_GRUB = 4, K_DMOLS = 5, K_GRUS = 6, K_GRUSB = 7, K_LSTMS = 8, K_LSTMSB = 9, K_LINSEQ = 10 };
enum DescFlag : int {
  DF_RELU = 1,         // K_LIN: leaky ReLU (f[0] = slope) on the result
  DF_A_PLAIN = 2,      // K_LIN: A is a row-major buffer written before the launch (ld[0]), not a polled T16 copy
  DF_ADD_POLLED = 4,   // K_LIN: `add` words are produced inside the launch
  DF_RM_SC1 = 8,       // the row-major output is polled word-wise by other workgroups
  DF_GENTLE = 16,      // off the critical path: nap between polls
  DF_CANARY = 32,      // one-word canary wait in front of the operand poll
  DF_A_SUM3 = 64,      // K_LIN: the polled operand is the sum of three slabs (p[0], p[8], p[9]) of partial sums
  DF_SEQ_GATE = 128,   // K_LINSEQ: the per-link auxiliary pointer is the derivative gate (backward chains), not the bias
};
constexpr int kMaxDesc = 24, kMaxPtr = 20;
struct Desc {
  int kind, ct, wg0, nwg, flags, K, s_begin, s_end;
  int ld[4];                      // leading dimensions of row-major operands (per kind, see pchain.hip)
  int n16[2];                     // 16-column chunks per row of the T16 outputs
  int i[4];                       // per kind
  float f[4];                     // per kind
  unsigned char sidx[kMaxPtr];    // stride-table index of every pointer
  const float* p[kMaxPtr];        // per kind
};
struct Program {
  int bf16 = 0;  // weights are bf16 T16 packs, products on the bf16 matrix pipe (see mgemm_trip)
  int rt_group = 1;  // row tiles per tile: 1 = pchain.h's 16-row tiles; 4 = row groups (pchain_rt.h; the VRNN tile kinds only, B > 128)
  int s_first = 0;  // the launch walks steps [s_first, S): a sequence may be cut into several launches (everything a later one needs is in the slabs)
  int ndesc, S, B, xcd;
  long stride[16];
  Ctl ctl;
  unsigned long long* prof;       // diagnostics (blvm_pchain_profile): ticks per descriptor of workgroups 0 and prof_wg
  int prof_wg, lds_products;      // lds_products: most products of any tile kind used (sizes the reduction scratch)
  Desc d[kMaxDesc];
};
// (the program reaches the device through pchain_resolve_kernel, a few descriptors per launch: kernel arguments are limited to 4 KB)

// host-side assembly of a program
struct Builder {
  Program p{};
  int nstride = 1;
  bool overflow = false;
  int stride_index(long v) {
    if (v == 0) return 0;
    for (int i = 1; i < nstride; ++i)
      if (p.stride[i] == v) return i;
    if (nstride >= 16) { overflow = true; return 0; }
    p.stride[nstride] = v;
    return nstride++;
  }
  // a descriptor of `ct` column tiles (x all row tiles) on workgroups [wg0, wg0 + nwg), active in steps [s_begin, s_end)
  Desc& add(int kind, int ct, int wg0, int nwg, int K, int flags, int s_begin, int s_end) {
    static Desc dummy;
    if (p.ndesc >= kMaxDesc) { overflow = true; return dummy; }  // (callers check `overflow`)
    Desc& d = p.d[p.ndesc++];
    d = Desc{};
    d.kind = kind; d.ct = ct; d.wg0 = wg0; d.nwg = nwg; d.K = K; d.flags = flags; d.s_begin = s_begin; d.s_end = s_end;
    return d;
  }
  void ptr(Desc& d, int k, const void* q, long stride = 0) {
    d.p[k] = static_cast<const float*>(q);
    d.sidx[k] = (unsigned char)(q ? stride_index(stride) : 0);
  }
};

// A run of n <= 4 consecutive links of one shape as ONE K_LINSEQ descriptor (pchain.hip): out_i = act(A_i W_i^T + bias_i) or, with
// `gated` (backward chains), (A_i W_i^T) masked by the derivative of the activation whose output is gate_i; A_0 = A16, A_i = o16 of
// link i-1.  Steps of the stepped pointers: a_step (A_0), aux_step (gates), rm_step[i] (row-major outputs), o16_step (T16 outputs).
struct SeqLink {
  const float* W;
  const float* aux;  // bias (not stepped) | gate (stepped by aux_step)
  float* orm;
  long rm_step;
  int ldo;
  float* o16;
};
// K0 / add0: the FIRST link may have its own K (its operand A16 is then [rows, K0]) and a row-major addend [B, ldadd0] stepping by
// add0_step -- the link in front of a run of same-shape links joins the run's visit (one descriptor walk less per step, ~1 us)
inline Desc& add_linseq(Builder& b, int ct, int wg0, int nwg, int K, bool relu, bool gated, int s_begin, int s_end, const float* A16, long a_step, int n,
                        const SeqLink* L, long aux_step, long o16_step, int n16, float slope, int ldgate, int K0 = 0, const float* add0 = nullptr,
                        long add0_step = 0, int ldadd0 = 0) {
  Desc& d = b.add(K_LINSEQ, ct, wg0, nwg, K, (relu ? DF_RELU : 0) | (gated ? DF_SEQ_GATE : 0), s_begin, s_end);
  b.ptr(d, 0, A16, a_step); b.ptr(d, 17, add0, add0_step);
  d.i[3] = (K0 != 0 && K0 != K) ? K0 : 0; d.i[0] = ldadd0;
  for (int i = 0; i < n && i < 4; ++i) {
    b.ptr(d, 1 + i, L[i].W); b.ptr(d, 5 + i, L[i].aux, gated ? aux_step : 0); b.ptr(d, 9 + i, L[i].orm, L[i].rm_step); b.ptr(d, 13 + i, L[i].o16, o16_step);
    d.ld[i] = L[i].ldo;
  }
  if (n > 4) b.overflow = true;
  d.n16[0] = n16; d.i[1] = n; d.i[2] = ldgate; d.f[0] = slope;
  return d;
}
// env BLVM_PCHAIN_LINSEQ=0: one descriptor per link (A/B switch)
// env BLVM_PCHAIN_MERGE=0: the link in front of a run keeps its own descriptor (A/B switch)
inline bool merge_first_enabled() {
  static const int v = [] { const char* e = getenv("BLVM_PCHAIN_MERGE"); return e ? atoi(e) : 1; }();
  return v != 0;
}
inline bool linseq_enabled() {
  static const int v = [] { const char* e = getenv("BLVM_PCHAIN_LINSEQ"); return e ? atoi(e) : 1; }();
  return v != 0;
}

// workgroups for `tiles` tiles out of `avail` (a multiple of 8, at least 8): XCD-aware placement deals ranges in eights
inline int range_for(int tiles, int avail) { return std::max(8, std::min(avail & ~7, (tiles + 7) & ~7)); }

}  // namespace pchain

// sentinel-fill `bytes` (a multiple of 4) at p (4-byte aligned) on `stream`: what hipMemsetAsync(p, 0xFF, bytes) does, as 16-byte
// stores from 2048 workgroups (the runtime's fill kernel runs 256 workgroups: 72 us for the VRNN backward slabs of [64,16000])
hipError_t pchain_fill_sentinel(void* p, size_t bytes, hipStream_t stream);
// enqueue the persistent launch of a program (pchain.hip); grid = highest workgroup any descriptor names
int pchain_launch(const pchain::Program& prog, hipStream_t stream);
// dst = T16 copy [ceil(B/16)*16, K] of the rows of src [B, K] (row stride ld; null: zeros); rows >= B are left alone (never read).
// n16 > 0: dst is a slab of n16 blocks per row tile (a concatenation; dst points at this part's first block)
int pchain_rows_to_t16(const float* src, int ld, int B, int K, float* dst, hipStream_t stream, int n16 = 0);
inline int device_cus() {
  static int v = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (const char* e = getenv("BLVM_PCHAIN_CUS")) {  // experiments: programs dealt over fewer workgroups than the chip has CUs
      const int m = atoi(e) & ~7;
      if (m >= 32 && m < n) n = m;
    }
    return n;
  }();
  return v;
}
constexpr int kPchainCarveMaxB = 128;  // the persistent kernels' extra buffers are carved for batches up to this size only
inline bool pchain_applies(int B) { return B <= pchain_max_batch() && B <= kPchainCarveMaxB; }
// the bf16-operand mode (common.h operand_bf16) of a sequence that runs as a persistent launch: its weights are packed as bf16
inline bool pchain_bf16(int B) { return operand_bf16() && pchain_applies(B) && device_cus() >= 32; }

}  // namespace blvm
