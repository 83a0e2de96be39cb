// pchain.hip — the persistent-chain engine: ONE kernel that executes a host-built program of link descriptors (pchain.h) for a
// whole recurrent sequence.  Replaces, per model, thousands of dependent launches (VRNN [64,16000]: 4 500 per train step) by one
// forward and one backward launch.  Pointer roles of a descriptor by tile kind:
//
//   K_LIN   p: 0 A (T16 copy, polled, ld[0] = its width when wider than K | row-major when DF_A_PLAIN, ld[0])  1 W (T16)  2 bias  3 add (ld[1])  4 gate (ld[2])
//              5 out row-major (ld[3])  6 out T16 (n16[0])  7 second out T16 (n16[1])  8, 9 (DF_A_SUM3) two more slabs of A: the
//              operand is the sum of the three            i: 0 width of W's packed rows when the product covers a K-range of them            f: 0 slope
//   K_HEAD  p: 0 P16  1 Q16  2 Wp  3 bp  4 Wq  5 bq  6 eps  7 mu_p  8 sd_p  9 mu_q  10 sd_q  11 raw_p  12 raw_q  13 muq_raw
//              14 z row-major (ld[3])  15 z T16 (n16[0])  16 second z T16 (n16[1])    i: 0 Z  1 residual      f: 0 beta  1 1/beta  2 sd_eps
//   K_GRU   p: 0 X16  1 Wih (T16)  2 xg  3 gh (polled words)  4 h_prev (polled words, ld[0])  5 h_new row-major (ld[3])
//              6 h_new T16 (n16[0])  7 rg  8 ug  9 ng  10 b_ih  11 second h_new T16 (n16[1])         i: 0 R
//   K_DZ    p: 0 D16  1 WT  2 D2_16  3 WT2  4 dz_add (ld[1])  5 mu_q  6 sd_q  7 mu_p  8 sd_p  9 eps  10 raw_q  11 raw_p  12 muq_raw
//              13 x_sl (int32)  14 c_raw  15 c_fn  16 dqh row-major  17 dqh T16  18 dph row-major  19 dph T16   (ld[3] = 2Z, n16[0])
//              i: 0 Z  1 residual  2 stride  3 t at s = 0 (t = i[3] - s)   f: 0 fn_floor  1 beta  2 sd_eps  3 first step with the product
//   K_GRUB  p: 0 D0_16  1 D1_16  2 W0  3 W1  4 g_in (polled words)  5 rg  6 ug  7 ng  8 gh  9 h_prev (ld[0])  10 dd (ld[0])
//              11 dgi row-major  12 dgi T16  13 dgh row-major  14 dgh T16  (ld[3] = 3R, n16[0])  15 ga  16 g_out  17 g_add (ld[1])
//              i: 0 R  1 first step with the products  2 first step WITHOUT gates  3 first step with g_in
//   K_DMOLS p: 0 dec (row-major, polled words, ld[0])  1 head W [F,F]  2 head b  3 u  4 v  5 x row-major (ld[3])  6 x T16 (n16[0])
//              i: 0 S  1 F  2 num_mix      f: 0 log_eps        (ct counts tiles of 4 samples)
//   K_LINSEQ  n = i[1] <= 4 CONSECUTIVE links of one shape (K, ct, range) walked inside one visit — out_i = act(A_i W_i^T + bias_i) with
//              A_0 = p[0] and A_i = the T16 output of link i-1: between them only the pointers change, so the walk of the program
//              (descriptor fetch, decode, tile list) is paid once per visit instead of once per link
//              p: 0 A_0  1+i W_i  5+i bias_i | gate_i (DF_SEQ_GATE, ld = i[2])  9+i out_i row-major (ld[i])  13+i out_i T16 (n16[0])
//              i: 1 n  2 ldgate      f: 0 slope
//   K_GRUS  p: 0 H16 (state entering the step)  1 Whh (T16)  2 b_hh  3 xg [T,B,3R] (time indexed)  4 lens  5 h_prev row-major
//              6 h_next row-major (ld[3])  7 h_next T16 (n16[0])  8 out (time indexed)  9 rg  10 ug  11 ng  12 ghn
//              i: 0 R  1 reverse  2 out_ts  3 out_ld                                   (recurrence step j = s)
//   K_GRUSB p: 0 DGH16 of step j+1  1 WhhT (T16)  2 dout (time indexed)  3 rg  4 ug  5 ng  6 ghn  7 h_prev  8 lens  9 G (in place)
//              10 DGI [T,B,3R] (time indexed)  11 DGH row-major (ld[3] = 3R)  12 DGH T16 (n16[0])  13 dh0
//              i: 0 R  1 reverse  2 out_ts  3 out_ld    n16[1]: T   (j = T-1-s; s = 0: no product; s = T: only dh0)
//   K_LSTMS p: 0 H16  1 Whh (T16)  2 b_hh  3 xg of the step [B,4H]  4 lens  5 h_prev  6 h_next row-major (ld[3])  7 h_next T16
//              8 c_prev  9 c_next  10 out  11 gates [B,4H]        i: 0 H                (t = s)
//   K_LSTMSB p: 0 DG16 of step t+1  1 WhhT (T16)  2 dout  3 gates  4 c_s (c_{s+1} = one [B,H] slab further)  5 DC (in place)
//              6 DG row-major (ld[3] = 4H)  7 DG T16 (n16[0])  8 dh0       i: 0 H       n16[1]: T   (t = T-1-s)
#include <algorithm>
#include <mutex>

#include "common.h"
#include "pchain.h"
#include "pchain_rt.h"

namespace blvm {
namespace {
using namespace pchain;

// A descriptor as the kernel reads it: 128 dwords, resolved (strides per pointer) by pchain_resolve_kernel right before the launch
// and copied into LDS by every workgroup.  Every wave keeps the descriptor of its current tile in TWO VGPRs (lane l holds dword l /
// 64 + l: two ds_read_b32) and reads a field with v_readlane; the NEXT descriptor's two LDS reads are issued at the top of the
// current tile.  LDS, not global memory: on gfx9 vector loads AND stores share the vmcnt counter, so waiting for a prefetched
// global load at the top of the next tile also sat out the previous tile's write-through stores (~0.9 us per tile, found with
// blvm_pchain_chain_probe against tools/pchain_probe.hip); LDS reads are counted by lgkmcnt.
//   dword 0 kind  1 ct  2 wg0  3 nwg  4 flags  5 K  6 s_begin  7 s_end  8..11 ld  12..13 n16  14..17 i  18..21 f
//         24 + 2k, 25 + 2k: pointer k (k < 20)        64 + k: its per-step stride in floats (int32)
//         84: number of tiles of THIS workgroup, 85 ..: its tiles as r0 | column tile << 16 — written once by each workgroup into
//         its own LDS copy (TileIter involves integer divisions: ~0.5 us per tile when run inside the step loop)
constexpr int kDescWords = 128;
enum { RD_KIND = 0, RD_CT, RD_WG0, RD_NWG, RD_FLAGS, RD_K, RD_SBEGIN, RD_SEND, RD_LD = 8, RD_N16 = 12, RD_I = 14, RD_F = 18, RD_P = 24, RD_ST = 64,
       RD_NT = 84, RD_TILE = 85 };
constexpr int kMaxTilesPerWg = kDescWords - RD_TILE;
struct Hdr {
  int ndesc, s0, S, B, xcd, prof_wg, lds_products;  // steps [s0, S)
  Ctl ctl;
  unsigned long long* prof;
};

// descriptors reach the device a few per launch (kernel arguments are limited to 4 KB)
constexpr int kResolveChunk = 12;
struct ProgramPart {
  int ndesc, first;
  long stride[16];
  Desc d[kResolveChunk];
};
static_assert(sizeof(ProgramPart) <= 4096, "kernel argument size");

__global__ void pchain_resolve_kernel(ProgramPart a, int* __restrict__ out) {
  out += (size_t)a.first * kDescWords;
  for (int e = threadIdx.x; e < a.ndesc * kDescWords; e += blockDim.x) {
    const int i = e / kDescWords, k = e % kDescWords;
    const Desc& d = a.d[i];
    int v = 0;
    if (k == RD_KIND) v = d.kind;
    else if (k == RD_CT) v = d.ct;
    else if (k == RD_WG0) v = d.wg0;
    else if (k == RD_NWG) v = d.nwg;
    else if (k == RD_FLAGS) v = d.flags;
    else if (k == RD_K) v = d.K;
    else if (k == RD_SBEGIN) v = d.s_begin;
    else if (k == RD_SEND) v = d.s_end;
    else if (k >= RD_LD && k < RD_LD + 4) v = d.ld[k - RD_LD];
    else if (k >= RD_N16 && k < RD_N16 + 2) v = d.n16[k - RD_N16];
    else if (k >= RD_I && k < RD_I + 4) v = d.i[k - RD_I];
    else if (k >= RD_F && k < RD_F + 4) v = __float_as_int(d.f[k - RD_F]);
    else if (k >= RD_P && k < RD_P + 2 * kMaxPtr) {
      const unsigned long long q = reinterpret_cast<unsigned long long>(d.p[(k - RD_P) >> 1]);
      v = (int)(((k - RD_P) & 1) ? (q >> 32) : (q & 0xffffffffull));
    } else if (k >= RD_ST && k < RD_ST + kMaxPtr) {
      v = (int)a.stride[d.sidx[k - RD_ST]];
    }
    out[e] = v;
  }
}

// the descriptor of the current tile, one dword per lane in two VGPRs, plus the pointers OF THE STEP it will be used in: lane k of
// (plo, phi) = pointer k + step * stride k, computed for all twenty pointers at once by four vector instructions when the
// descriptor is fetched (one tile ahead) — as scalar code in front of the tile it was ~7 dependent SALU instructions per pointer,
// ~0.3 us of every link
struct DescRegs {
  int v0, v1;
  unsigned plo, phi;
  __device__ __forceinline__ void fetch(const int* tab_lds, int i, int step) {
    const int lane = threadIdx.x & 63;
    const int* q = tab_lds + i * kDescWords;
    v0 = q[lane];
    v1 = q[64 + lane];
    const int k = lane < kMaxPtr ? lane : 0;
    const unsigned long long base = ((unsigned long long)(unsigned)q[RD_P + 2 * k + 1] << 32) | (unsigned)q[RD_P + 2 * k];
    const unsigned long long p = base + (unsigned long long)((long long)step * (long long)q[RD_ST + k] * 4);
    plo = (unsigned)p;
    phi = (unsigned)(p >> 32);
  }
  template <int IDX>
  __device__ __forceinline__ int w() const {
    return IDX < 64 ? __builtin_amdgcn_readlane(v0, IDX & 63) : __builtin_amdgcn_readlane(v1, IDX & 63);
  }
  __device__ __forceinline__ int tile(int k) const { return __builtin_amdgcn_readlane(v1, RD_TILE - 64 + k); }  // k wave-uniform
  template <int IDX>
  __device__ __forceinline__ float f() const { return __int_as_float(w<RD_F + IDX>()); }
  template <int K>
  __device__ __forceinline__ const float* base() const {
    const unsigned long long lo = (unsigned)w<RD_P + 2 * K>(), hi = (unsigned)w<RD_P + 2 * K + 1>();
    return reinterpret_cast<const float*>((hi << 32) | lo);
  }
  // pointer k at the step the descriptor was fetched for (null stays null: its stride is 0)
  template <int K>
  __device__ __forceinline__ const float* p(int) const {
    const unsigned long long lo = (unsigned)__builtin_amdgcn_readlane((int)plo, K), hi = (unsigned)__builtin_amdgcn_readlane((int)phi, K);
    return reinterpret_cast<const float*>((hi << 32) | lo);
  }
  template <int K>
  __device__ __forceinline__ float* m(int s) const { return const_cast<float*>(p<K>(s)); }
  // the same with a wave-uniform RUN-TIME pointer index (v_readlane takes the lane from an SGPR)
  __device__ __forceinline__ const float* pdyn(int k) const {
    const unsigned long long lo = (unsigned)__builtin_amdgcn_readlane((int)plo, k), hi = (unsigned)__builtin_amdgcn_readlane((int)phi, k);
    return reinterpret_cast<const float*>((hi << 32) | lo);
  }
  __device__ __forceinline__ const float* basedyn(int k) const {  // (RD_P + 2k < 64: pointers live in v0)
    const unsigned long long lo = (unsigned)__builtin_amdgcn_readlane(v0, RD_P + 2 * k), hi = (unsigned)__builtin_amdgcn_readlane(v0, RD_P + 2 * k + 1);
    return reinterpret_cast<const float*>((hi << 32) | lo);
  }
  __device__ __forceinline__ int wdyn(int idx) const { return __builtin_amdgcn_readlane(v0, idx); }  // idx < 64
};

template <int NW, bool BF>
__global__ __launch_bounds__(NW * 64, 1) void pchain_kernel(const int* __restrict__ tab, Hdr a) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];  // [descriptors | profile | 2 x (lds_products x NW x 256) floats]
  int* const ltab = reinterpret_cast<int*>(lds_all);
  unsigned long long* const pacc = reinterpret_cast<unsigned long long*>(lds_all + sizeof(int) * kDescWords * kMaxDesc);
  float* const red0 = reinterpret_cast<float*>(lds_all + sizeof(int) * kDescWords * kMaxDesc + 32 * sizeof(unsigned long long));
  float* const red1 = red0 + a.lds_products * NW * 256;
  const int w = blockIdx.x, B = a.B, rt = (B + 15) / 16;
  const bool xcd = a.xcd != 0;
  for (int e = threadIdx.x; e < a.ndesc * kDescWords; e += NW * 64) ltab[e] = tab[e];
  if (threadIdx.x < 32) pacc[threadIdx.x] = 0ull;
  unsigned mine = 0;  // the descriptors this workgroup has tiles of
  for (int i = 0; i < a.ndesc; ++i) {
    const int* q = tab + (size_t)i * kDescWords;
    if (TileIter(w, q[RD_WG0], q[RD_NWG], rt, q[RD_CT], xcd).valid()) mine |= 1u << i;
  }
  __syncthreads();
  if (threadIdx.x < a.ndesc) {  // this workgroup's tiles of descriptor threadIdx.x, into its own copy of the table
    int* q = ltab + threadIdx.x * kDescWords;
    int k = 0;
    for (TileIter it(w, q[RD_WG0], q[RD_NWG], rt, q[RD_CT], xcd); it.valid() && k < kMaxTilesPerWg; it.next()) q[RD_TILE + k++] = it.r0() | (it.c() << 16);
    q[RD_NT] = k;
  }
  __syncthreads();
  if (mine == 0) return;
  int par = 0;
  auto red = [&]() { par ^= 1; return par ? red0 : red1; };
  Poll pl{a.ctl, 0u, false, 1};
  // per-descriptor tick counters (blvm_pchain_profile) only in -DPCHAIN_PROF builds (and the stamped -DPCHAIN_TPROF* ones): in the
  // product kernel their state would sit in registers the walk needs (tools/probe_engine_chain.py: every live scalar costs)
#if defined(PCHAIN_PROF) || defined(PCHAIN_TPROF) || defined(PCHAIN_TPROF2)
  const bool profiled = a.prof != nullptr && (w == 0 || w == a.prof_wg) && threadIdx.x == 0;
#else
  constexpr bool profiled = false;
#endif
  unsigned long long tprev = profiled ? wall_clock64() : 0ull;
#ifdef PCHAIN_TPROF2
  unsigned long long tq[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tq_end = 0ull;
#endif
  DescRegs d, nx;
  d.fetch(ltab, __builtin_ctz(mine), a.s0);
  for (int s = a.s0; s < a.S; ++s) {
    for (unsigned m = mine; m != 0; m &= m - 1) {
      const int i = __builtin_ctz(m);
#ifdef PCHAIN_TPROF2
      const unsigned long long tq0 = wall_clock64();
      if (tq_end) tq[2] += tq0 - tq_end;
#endif
      // request the next descriptor of this workgroup's walk now; it lands while this tile runs
      const unsigned rest = m & (m - 1);
      const int nx_i = __builtin_ctz(rest != 0 ? rest : mine), nx_s = rest != 0 ? s : s + 1;
      const int kind = d.w<RD_KIND>(), flags = d.w<RD_FLAGS>(), K = d.w<RD_K>();
      const bool active = s >= d.w<RD_SBEGIN>() && s < d.w<RD_SEND>();
#ifdef PCHAIN_TPROF2
      const unsigned long long tqa = wall_clock64();
      tq[3] += tqa - tq0;
#endif
      // (a linear tile fetches the next descriptor itself, in the shadow of its operand wait)
      if (!active || (kind != K_LIN && kind != K_LINSEQ)) nx.fetch(ltab, nx_i, nx_s);
      if (active) {
        const int nt = d.w<RD_NT>();
        pl.nap = (flags & DF_GENTLE) ? 16 : 1;
        pl.code = ((unsigned)s << 4) | (unsigned)i;
        switch (kind) {
          case K_LIN: {
            // Only what the operand loads need is taken out of the descriptor here; everything the epilogue needs (and the next
            // descriptor's fetch) is done by `late`, which the tile calls once its first loads are in flight.
            const bool a_polled = !(flags & DF_A_PLAIN);
            const float *A = d.p<0>(s), *W = d.base<1>();
            const float *A2 = (flags & DF_A_SUM3) ? d.p<8>(s) : nullptr, *A3 = (flags & DF_A_SUM3) ? d.p<9>(s) : nullptr;
            const int ld0 = d.w<RD_LD + 0>(), w_width = d.w<RD_I + 0>();
#ifdef PCHAIN_TPROF2
            const unsigned long long tqb = wall_clock64();
            tq[4] += tqb - tqa;
#endif
            auto late = [&]() {
              __builtin_amdgcn_sched_barrier(0);
              nx.fetch(ltab, nx_i, nx_s);  // (per tile: the same lanes again — cheaper than a flag carried through the visit)
              return LinLate{d.base<2>(), d.p<3>(s), d.p<4>(s), d.w<RD_LD + 1>(), d.w<RD_LD + 2>(), (flags & DF_ADD_POLLED) != 0, (flags & DF_RELU) != 0, d.f<0>(),
                             Out{d.m<5>(s), d.w<RD_LD + 3>(), (flags & DF_RM_SC1) != 0, d.m<6>(s), d.w<RD_N16>(), d.m<7>(s), d.w<RD_N16 + 1>()}};
            };
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk), tr0 = trc & 0xffff, tc0 = (trc >> 16) * 16;
              if ((flags & DF_CANARY) && a_polled) canary_wait(A, tr0, K, pl, ld0);
#ifdef PCHAIN_TPROF2
              const unsigned long long tq1 = wall_clock64();
              tq[0] += tq1 - tq0;
#endif
              tile_lin_late<NW, BF>(A, ld0, a_polled, W, K, late, tr0, tc0, B, red(), pl, A2, A3, w_width);
#ifdef PCHAIN_TPROF2
              tq_end = wall_clock64();
              tq[1] += tq_end - tq1;
#endif
            }
            if (nt == 0) nx.fetch(ltab, nx_i, nx_s);  // (a visit without tiles)
          } break;
          case K_LINSEQ: {
            const int n = d.w<RD_I + 1>();
            const bool gated = (flags & DF_SEQ_GATE) != 0;
            const float* A = d.p<0>(s);
            const int K0 = d.w<RD_I + 3>();  // the FIRST link's K when it is not the run's (0: it is): the link in front of a run joins its visit
            for (int lq = 0; lq < n; ++lq) {  // (links outside, tiles inside: link li+1 of any tile needs link li of ALL tiles of its row tile)
              int li = lq;
              asm volatile("" : "+s"(li));  // opaque: the five lane indices below are li + constant where they are used, not five more loop counters in SGPRs
              pl.code = ((unsigned)s << 4) | (unsigned)i | ((unsigned)li << 28);
              const float* W = d.basedyn(1 + li);
              const int Kl = (li == 0 && K0 != 0) ? K0 : K;
              auto late = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                if (li == 0) nx.fetch(ltab, nx_i, nx_s);  // the next descriptor, once per visit, behind the first link's operand loads (every tile of it: the same lanes again)
                const float* aux = d.pdyn(5 + li);  // bias (stride 0: the stepped pointer IS the base) | gate (stepped)
                const float* add0 = li == 0 ? d.p<17>(s) : nullptr;  // the first link's row-major addend [B, i[0]] (null: none)
                return LinLate{gated ? nullptr : aux, add0, gated ? aux : nullptr, d.w<RD_I + 0>(), d.w<RD_I + 2>(), false, (flags & DF_RELU) != 0, d.f<0>(),
                               Out{const_cast<float*>(d.pdyn(9 + li)), d.wdyn(RD_LD + li), false, const_cast<float*>(d.pdyn(13 + li)), d.w<RD_N16>()}};
              };
              // (a run's links are plain: a partial-sum operand or a K-range is a K_LIN in front of the run.  The summing link as a run's first
              // link was built and measured, r03: the three-slab product inlined at this call site costs the WHOLE kernel +0.4 ms per
              // VRNN step, used or not, and saves 0.08)
              for (int tk = 0; tk < nt; ++tk) {
                const int trc = d.tile(tk), tr0 = trc & 0xffff, tc0 = (trc >> 16) * 16;
                tile_lin_late<NW, BF>(A, 0, true, W, Kl, late, tr0, tc0, B, red(), pl);
              }
              A = d.pdyn(13 + li);  // the next link multiplies what this one stored
            }
            if (nt == 0) nx.fetch(ltab, nx_i, nx_s);  // (a visit without tiles)
          } break;
#ifndef PCHAIN_ONLY_LIN
          case K_HEAD: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const HeadOut o{d.m<7>(s), d.m<8>(s), d.m<9>(s), d.m<10>(s), d.m<11>(s), d.m<12>(s), d.m<13>(s),
                            Out{d.m<14>(s), ld3, false, d.m<15>(s), n16, d.m<16>(s), d.w<RD_N16 + 1>()}};
            const int Z = d.w<RD_I + 0>(), residual = d.w<RD_I + 1>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_head<NW, BF>(d.p<0>(s), d.p<1>(s), true, d.base<2>(), d.base<3>(), d.base<4>(), d.base<5>(), d.p<6>(s), o, K, Z, residual, d.f<0>(), d.f<1>(),
                            d.f<2>(), trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_GRU: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const Out o{d.m<5>(s), ld3, true, d.m<6>(s), n16, d.m<11>(s), d.w<RD_N16 + 1>()};
            const int R = d.w<RD_I + 0>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_gru<NW, BF>(d.p<0>(s), 0, true, d.base<1>(), K, d.p<2>(s), d.base<10>(), d.p<3>(s), d.p<4>(s), ld0, R, o, d.m<7>(s), d.m<8>(s), d.m<9>(s),
                           trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_DZ: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            DzIn z;
            z.mu_q = d.p<5>(s); z.sd_q = d.p<6>(s); z.mu_p = d.p<7>(s); z.sd_p = d.p<8>(s); z.eps = d.p<9>(s); z.raw_q = d.p<10>(s); z.raw_p = d.p<11>(s);
            z.muq_raw = d.p<12>(s);
            z.x_sl = reinterpret_cast<const int32_t*>(d.base<13>()); z.c_raw = d.base<14>(); z.c_fn = d.base<15>();
            z.t = d.w<RD_I + 3>() - s; z.stride = d.w<RD_I + 2>(); z.residual = d.w<RD_I + 1>();
            z.fn_floor = d.f<0>(); z.beta = d.f<1>(); z.sd_eps = d.f<2>();
            z.has_gemm = s >= (int)d.f<3>();
            const int Z = d.w<RD_I + 0>();
            const Out oq{d.m<16>(s), ld3, false, d.m<17>(s), n16}, op{d.m<18>(s), ld3, false, d.m<19>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_dz<NW, BF>(d.p<0>(s), d.base<1>(), d.p<2>(s), d.base<3>(), true, d.p<4>(s), ld1, (flags & DF_ADD_POLLED) != 0, z, oq, op, K, Z, trc & 0xffff,
                          (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_GRUB: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            GrubIn g;
            g.D0 = d.p<0>(s); g.D1 = d.p<1>(s); g.W0 = d.base<2>(); g.W1 = d.base<3>(); g.g_in = d.p<4>(s); g.g_add = d.p<17>(s); g.ld_gadd = ld1;
            g.rg = d.p<5>(s); g.ug = d.p<6>(s); g.ng = d.p<7>(s); g.gh = d.p<8>(s); g.hprev = d.p<9>(s); g.dd = d.p<10>(s); g.ldh = ld0;
            g.dgi = Out{d.m<11>(s), ld3, false, d.m<12>(s), n16};
            g.dgh = Out{d.m<13>(s), ld3, false, d.m<14>(s), n16};
            g.ga = d.m<15>(s); g.g_out = const_cast<float*>(d.base<16>());
            g.has_gemm = s >= d.w<RD_I + 1>(); g.has_gates = s < d.w<RD_I + 2>(); g.has_gin = s >= d.w<RD_I + 3>();
            const int R = d.w<RD_I + 0>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_grub<NW, BF>(g, K, R, trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_GRUS: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const int R = d.w<RD_I + 0>();
            GruSeqIn g;
            g.H16 = d.p<0>(s); g.Whh = d.base<1>(); g.bhh = d.base<2>(); g.xg = d.base<3>(); g.lens = reinterpret_cast<const int32_t*>(d.base<4>());
            g.hprev = d.p<5>(s); g.out = const_cast<float*>(d.base<8>()); g.rg = d.m<9>(s); g.ug = d.m<10>(s); g.ng = d.m<11>(s); g.ghn = d.m<12>(s);
            g.out_ts = d.w<RD_I + 2>(); g.out_ld = d.w<RD_I + 3>(); g.j = s; g.reverse = d.w<RD_I + 1>();
            const Out o{d.m<6>(s), ld3, false, d.m<7>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_gru_seq<NW, BF>(g, o, R, trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_GRUSB: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const int R = d.w<RD_I + 0>(), T = d.w<RD_N16 + 1>();
            GruSeqBwdIn g;
            g.DGHn16 = d.p<0>(s); g.WhhT = d.base<1>(); g.dout = d.base<2>(); g.rg = d.p<3>(s); g.ug = d.p<4>(s); g.ng = d.p<5>(s); g.ghn = d.p<6>(s);
            g.hprev = d.p<7>(s); g.lens = reinterpret_cast<const int32_t*>(d.base<8>()); g.G = const_cast<float*>(d.base<9>());
            g.DGI = const_cast<float*>(d.base<10>()); g.dh0 = const_cast<float*>(d.base<13>());
            g.out_ts = d.w<RD_I + 2>(); g.out_ld = d.w<RD_I + 3>(); g.j = T - 1 - s; g.reverse = d.w<RD_I + 1>();
            g.has_gemm = s >= 1; g.has_gates = s < T;
            const Out o{d.m<11>(s), ld3, false, d.m<12>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_gru_seq_bwd<NW, BF>(g, o, R, trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_LSTMS: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const int H = d.w<RD_I + 0>();
            LstmSeqIn g;
            g.H16 = d.p<0>(s); g.Whh = d.base<1>(); g.bhh = d.base<2>(); g.xg = d.p<3>(s); g.lens = reinterpret_cast<const int32_t*>(d.base<4>());
            g.hprev = d.p<5>(s); g.cprev = d.p<8>(s); g.cnext = d.m<9>(s); g.out = d.m<10>(s); g.gates = d.m<11>(s); g.t = s;
            const Out o{d.m<6>(s), ld3, false, d.m<7>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_lstm_seq<NW, BF>(g, o, H, trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_LSTMSB: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const int H = d.w<RD_I + 0>(), T = d.w<RD_N16 + 1>();
            LstmSeqBwdIn g;
            g.DGn16 = d.p<0>(s); g.WhhT = d.base<1>(); g.dout = d.p<2>(s); g.gates = d.p<3>(s); g.c_s = d.p<4>(s); g.c_s1 = g.c_s + (size_t)B * H;
            g.DC = const_cast<float*>(d.base<5>()); g.dh0 = const_cast<float*>(d.base<8>());
            g.has_gemm = s >= 1; g.has_gates = s < T;
            const Out o{d.m<6>(s), ld3, false, d.m<7>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_lstm_seq_bwd<NW, BF>(g, o, H, trc & 0xffff, (trc >> 16) * 16, B, red(), pl);
            }
          } break;
          case K_DMOLS: {
            const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld2 = d.w<RD_LD + 2>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
            (void)ld0; (void)ld1; (void)ld2; (void)ld3; (void)n16;
            const Out o{d.m<5>(s), ld3, false, d.m<6>(s), n16};
            const int S = d.w<RD_I + 0>(), F = d.w<RD_I + 1>(), nmix = d.w<RD_I + 2>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_dmol_sample<NW>(d.p<0>(s), ld0, d.base<1>(), d.base<2>(), d.p<3>(s), d.p<4>(s), S, F, nmix, d.f<0>(), o, trc & 0xffff, (trc >> 16) * 4, B,
                                   red0, pl);
            }
          } break;
#endif
          default: break;
        }
        if (profiled) {
          const unsigned long long now = wall_clock64();
          pacc[i] += now - tprev;
          tprev = now;
        }
      }
      d = nx;
    }
  }
  if (profiled) {
    for (int i = 0; i < a.ndesc; ++i) a.prof[(w == 0 ? 0 : 32) + i] += pacc[i];
#ifdef PCHAIN_TPROF
    if (w == 0) for (int l = 0; l < 8; ++l) a.prof[56 + l] += pl.tp[l];
#endif
#ifdef PCHAIN_TPROF2
    if (w == 0) for (int l = 0; l < 3; ++l) a.prof[60 + l] += tq[l];
    if (w == 0) for (int l = 3; l < 6; ++l) a.prof[70 + l] += tq[l];
#endif
  }
}

// The same walk on ROW GROUPS (pchain_rt.h): a tile is up to RT row tiles of one column tile, the program's tile lists count
// groups (rt = row groups here), and only the tile kinds of the VRNN programs exist.  A separate kernel on purpose: the 16-row
// kernel above is sensitive to every live scalar (tools/probe_engine_chain.py) and must not change when this one does.
template <int NW, bool BF, int RT>
__global__ __launch_bounds__(NW * 64, 1) void pchain_rt_kernel(const int* __restrict__ tab, Hdr a) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  int* const ltab = reinterpret_cast<int*>(lds_all);
  float* const red0 = reinterpret_cast<float*>(lds_all + sizeof(int) * kDescWords * kMaxDesc + 32 * sizeof(unsigned long long));
  float* const red1 = red0 + a.lds_products * NW * 256;
  const int w = blockIdx.x, B = a.B, rt = ((B + 15) / 16 + RT - 1) / RT;  // row groups
  const bool xcd = a.xcd != 0;
  for (int e = threadIdx.x; e < a.ndesc * kDescWords; e += NW * 64) ltab[e] = tab[e];
  unsigned mine = 0;
  for (int i = 0; i < a.ndesc; ++i) {
    const int* q = tab + (size_t)i * kDescWords;
    if (TileIter(w, q[RD_WG0], q[RD_NWG], rt, q[RD_CT], xcd).valid()) mine |= 1u << i;
  }
  __syncthreads();
  if (threadIdx.x < a.ndesc) {
    int* q = ltab + threadIdx.x * kDescWords;
    int k = 0;
    for (TileIter it(w, q[RD_WG0], q[RD_NWG], rt, q[RD_CT], xcd); it.valid() && k < kMaxTilesPerWg; it.next()) q[RD_TILE + k++] = (it.r0() * RT) | (it.c() << 16);
    q[RD_NT] = k;
  }
  __syncthreads();
  if (mine == 0) return;
  int par = 0;
  auto red = [&]() { par ^= 1; return par ? red0 : red1; };
  Poll pl{a.ctl, 0u, false, 1};
  DescRegs d, nx;
  d.fetch(ltab, __builtin_ctz(mine), a.s0);
  for (int s = a.s0; s < a.S; ++s) {
    for (unsigned m = mine; m != 0; m &= m - 1) {
      const int i = __builtin_ctz(m);
      const unsigned rest = m & (m - 1);
      const int nx_i = __builtin_ctz(rest != 0 ? rest : mine), nx_s = rest != 0 ? s : s + 1;
      const int kind = d.w<RD_KIND>(), flags = d.w<RD_FLAGS>(), K = d.w<RD_K>();
      const bool active = s >= d.w<RD_SBEGIN>() && s < d.w<RD_SEND>();
      nx.fetch(ltab, nx_i, nx_s);
      if (active) {
        const int nt = d.w<RD_NT>();
        pl.nap = (flags & DF_GENTLE) ? 16 : 1;
        pl.code = ((unsigned)s << 4) | (unsigned)i;
        const int ld0 = d.w<RD_LD + 0>(), ld1 = d.w<RD_LD + 1>(), ld3 = d.w<RD_LD + 3>(), n16 = d.w<RD_N16>();
        switch (kind) {
          case K_LIN: {
            const bool a_polled = !(flags & DF_A_PLAIN);
            const float *A = d.p<0>(s), *W = d.base<1>();
            const float *A2 = (flags & DF_A_SUM3) ? d.p<8>(s) : nullptr, *A3 = (flags & DF_A_SUM3) ? d.p<9>(s) : nullptr;
            const int w_width = d.w<RD_I + 0>();
            auto late = [&]() {
              return LinLate{d.base<2>(), d.p<3>(s), d.p<4>(s), ld1, d.w<RD_LD + 2>(), (flags & DF_ADD_POLLED) != 0, (flags & DF_RELU) != 0, d.f<0>(),
                             Out{d.m<5>(s), ld3, (flags & DF_RM_SC1) != 0, d.m<6>(s), n16, d.m<7>(s), d.w<RD_N16 + 1>()}};
            };
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk), tr0 = trc & 0xffff, tc0 = (trc >> 16) * 16;
              if ((flags & DF_CANARY) && a_polled) {
                for (int q = 0; q < RT && tr0 + 16 * q < B; ++q) canary_wait(A, tr0 + 16 * q, K, pl, ld0);
              }
              tile_lin_rt<NW, BF, RT, 1>(A, ld0, a_polled, W, K, late, tr0, tc0, B, red, pl, A2, A3, w_width);
            }
          } break;
          case K_LINSEQ: {
            const int n = d.w<RD_I + 1>();
            const bool gated = (flags & DF_SEQ_GATE) != 0;
            const float* A = d.p<0>(s);
            const int K0 = d.w<RD_I + 3>();  // the first link's own K (0: the run's), as in pchain_kernel
            for (int li = 0; li < n; ++li) {
              pl.code = ((unsigned)s << 4) | (unsigned)i | ((unsigned)li << 28);
              const float* W = d.basedyn(1 + li);
              const int Kl = (li == 0 && K0 != 0) ? K0 : K;
              auto late = [&]() {
                const float* aux = d.pdyn(5 + li);
                const float* add0 = li == 0 ? d.p<17>(s) : nullptr;
                return LinLate{gated ? nullptr : aux, add0, gated ? aux : nullptr, d.w<RD_I + 0>(), d.w<RD_I + 2>(), false, (flags & DF_RELU) != 0, d.f<0>(),
                               Out{const_cast<float*>(d.pdyn(9 + li)), d.wdyn(RD_LD + li), false, const_cast<float*>(d.pdyn(13 + li)), n16}};
              };
              for (int tk = 0; tk < nt; ++tk) {
                const int trc = d.tile(tk), tr0 = trc & 0xffff, tc0 = (trc >> 16) * 16;
                tile_lin_rt<NW, BF, RT, 1>(A, 0, true, W, Kl, late, tr0, tc0, B, red, pl);
              }
              A = d.pdyn(13 + li);
            }
          } break;
          case K_HEAD: {
            const HeadOut o{d.m<7>(s), d.m<8>(s), d.m<9>(s), d.m<10>(s), d.m<11>(s), d.m<12>(s), d.m<13>(s),
                            Out{d.m<14>(s), ld3, false, d.m<15>(s), n16, d.m<16>(s), d.w<RD_N16 + 1>()}};
            const int Z = d.w<RD_I + 0>(), residual = d.w<RD_I + 1>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_head_rt<NW, BF, RT>(d.p<0>(s), d.p<1>(s), true, d.base<2>(), d.base<3>(), d.base<4>(), d.base<5>(), d.p<6>(s), o, K, Z, residual, d.f<0>(),
                                       d.f<1>(), d.f<2>(), trc & 0xffff, (trc >> 16) * 16, B, red, pl);
            }
          } break;
          case K_GRU: {
            const Out o{d.m<5>(s), ld3, true, d.m<6>(s), n16, d.m<11>(s), d.w<RD_N16 + 1>()};
            const int R = d.w<RD_I + 0>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_gru_rt<NW, BF, RT>(d.p<0>(s), 0, true, d.base<1>(), K, d.p<2>(s), d.base<10>(), d.p<3>(s), d.p<4>(s), ld0, R, o, d.m<7>(s), d.m<8>(s),
                                      d.m<9>(s), trc & 0xffff, (trc >> 16) * 16, B, red, pl);
            }
          } break;
          case K_DZ: {
            DzIn z;
            z.mu_q = d.p<5>(s); z.sd_q = d.p<6>(s); z.mu_p = d.p<7>(s); z.sd_p = d.p<8>(s); z.eps = d.p<9>(s); z.raw_q = d.p<10>(s); z.raw_p = d.p<11>(s);
            z.muq_raw = d.p<12>(s);
            z.x_sl = reinterpret_cast<const int32_t*>(d.base<13>()); z.c_raw = d.base<14>(); z.c_fn = d.base<15>();
            z.t = d.w<RD_I + 3>() - s; z.stride = d.w<RD_I + 2>(); z.residual = d.w<RD_I + 1>();
            z.fn_floor = d.f<0>(); z.beta = d.f<1>(); z.sd_eps = d.f<2>();
            z.has_gemm = s >= (int)d.f<3>();
            const int Z = d.w<RD_I + 0>();
            const Out oq{d.m<16>(s), ld3, false, d.m<17>(s), n16}, op{d.m<18>(s), ld3, false, d.m<19>(s), n16};
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_dz_rt<NW, BF, RT>(d.p<0>(s), d.base<1>(), true, d.p<4>(s), ld1, (flags & DF_ADD_POLLED) != 0, z, oq, op, K, Z, trc & 0xffff, (trc >> 16) * 16, B,
                                     red, pl);
            }
          } break;
          case K_GRUB: {
            GrubIn g;
            g.D0 = d.p<0>(s); g.D1 = d.p<1>(s); g.W0 = d.base<2>(); g.W1 = d.base<3>(); g.g_in = d.p<4>(s); g.g_add = d.p<17>(s); g.ld_gadd = ld1;
            g.rg = d.p<5>(s); g.ug = d.p<6>(s); g.ng = d.p<7>(s); g.gh = d.p<8>(s); g.hprev = d.p<9>(s); g.dd = d.p<10>(s); g.ldh = ld0;
            g.dgi = Out{d.m<11>(s), ld3, false, d.m<12>(s), n16};
            g.dgh = Out{d.m<13>(s), ld3, false, d.m<14>(s), n16};
            g.ga = d.m<15>(s); g.g_out = const_cast<float*>(d.base<16>());
            g.has_gemm = s >= d.w<RD_I + 1>(); g.has_gates = s < d.w<RD_I + 2>(); g.has_gin = s >= d.w<RD_I + 3>();
            const int R = d.w<RD_I + 0>();
            for (int tk = 0; tk < nt; ++tk) {
              const int trc = d.tile(tk);
              tile_grub_rt<NW, BF, RT>(g, K, R, trc & 0xffff, (trc >> 16) * 16, B, red, pl);
            }
          } break;
          default: break;
        }
      }
      d = nx;
    }
  }
}

__global__ void rows_to_t16_kernel(const float* src, int ld, int B, int K, float* dst, int n16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int row = i / K, k = i % K;
  dst[((size_t)(row >> 4) * n16 + (k >> 4)) * 256 + ((row & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)] = src ? src[(size_t)row * ld + k] : 0.f;
}

}  // namespace

int pchain_rows_to_t16(const float* src, int ld, int B, int K, float* dst, hipStream_t stream, int n16) {
  BLVM_REQUIRE(B > 0 && K > 0 && K % 16 == 0 && dst != nullptr && (n16 == 0 || n16 >= K / 16), "pchain_rows_to_t16: bad arguments");
  hipLaunchKernelGGL(rows_to_t16_kernel, dim3((B * K + 255) / 256), dim3(256), 0, stream, src, ld, B, K, dst, n16 > 0 ? n16 : K / 16);
  BLVM_CHECK_LAUNCH("pchain_rows_to_t16");
  return BLVM_OK;
}

namespace {
__global__ __launch_bounds__(256) void fill_sentinel_kernel(unsigned* __restrict__ p, size_t n) {
  const size_t head = std::min<size_t>(n, ((16 - (reinterpret_cast<unsigned long long>(p) & 15)) & 15) / 4);  // words in front of the first 16-byte boundary
  uint4* const v = reinterpret_cast<uint4*>(p + head);
  const size_t nv = (n - head) / 4;
  const uint4 ff = make_uint4(pchain::SENTINEL, pchain::SENTINEL, pchain::SENTINEL, pchain::SENTINEL);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) v[i] = ff;
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) p[threadIdx.x] = pchain::SENTINEL;
    const size_t t0 = head + nv * 4;
    if (t0 + threadIdx.x < n) p[t0 + threadIdx.x] = pchain::SENTINEL;
  }
}
}  // namespace

hipError_t pchain_fill_sentinel(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if ((reinterpret_cast<unsigned long long>(p) & 3) != 0 || (bytes & 3) != 0) return hipMemsetAsync(p, 0xFF, bytes, stream);
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(fill_sentinel_kernel, dim3(grid), dim3(256), 0, stream, static_cast<unsigned*>(p), n);
  return hipGetLastError();
}

int pchain_launch(const pchain::Program& prog, hipStream_t stream) {
  BLVM_REQUIRE(prog.ndesc > 0 && prog.ndesc <= pchain::kMaxDesc && prog.S > 0 && prog.B > 0, "pchain: bad program (%d descriptors, %d steps)", prog.ndesc, prog.S);
  int grid = 0;
  for (int i = 0; i < prog.ndesc; ++i) {
    const pchain::Desc& d = prog.d[i];
    BLVM_REQUIRE(d.nwg > 0 && d.wg0 >= 0 && d.ct > 0 && d.K > 0 && d.K % 16 == 0, "pchain: bad descriptor %d", i);
    BLVM_REQUIRE(!prog.xcd || d.nwg % 8 == 0, "pchain: XCD-aware placement needs ranges of 8 k workgroups (descriptor %d has %d)", i, d.nwg);
    grid = std::max(grid, d.wg0 + d.nwg);
    const int rt = ((prog.B + 15) / 16 + prog.rt_group - 1) / prog.rt_group;  // row tiles, or row groups
    const int per_wg = prog.xcd ? (((d.ct + 7) / 8) * rt + d.nwg / 8 - 1) / (d.nwg / 8) : (d.ct * rt + d.nwg - 1) / d.nwg;
    BLVM_REQUIRE(per_wg <= kMaxTilesPerWg, "pchain: descriptor %d gives a workgroup %d tiles (at most %d)", i, per_wg, kMaxTilesPerWg);
  }
  int dev = 0, cus = 0;
  BLVM_HIP(hipGetDevice(&dev));
  BLVM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  BLVM_REQUIRE(grid <= cus, "pchain: the program names %d workgroups, the device has %d CUs (every workgroup must be resident)", grid, cus);
  // the resolved table: a slot of a small ring in library-owned device memory (launches on one stream are ordered; the ring keeps
  // launches that overlap on different streams apart)
  int* tab = nullptr;
  {
    static std::mutex mu;
    static int tab_dev = -1;
    static int* ring = nullptr;
    static unsigned next = 0;
    constexpr unsigned kSlots = 8;
    std::lock_guard<std::mutex> lock(mu);
    if (tab_dev != dev) {
      BLVM_HIP(hipMalloc(reinterpret_cast<void**>(&ring), sizeof(int) * kDescWords * pchain::kMaxDesc * kSlots));
      tab_dev = dev;
    }
    tab = ring + (size_t)(next++ % kSlots) * kDescWords * pchain::kMaxDesc;
  }
  for (int first = 0; first < prog.ndesc; first += kResolveChunk) {
    ProgramPart part;
    part.ndesc = std::min(kResolveChunk, prog.ndesc - first);
    part.first = first;
    for (int k = 0; k < 16; ++k) part.stride[k] = prog.stride[k];
    for (int k = 0; k < part.ndesc; ++k) part.d[k] = prog.d[first + k];
    hipLaunchKernelGGL(pchain_resolve_kernel, dim3(1), dim3(256), 0, stream, part, tab);
  }
  BLVM_REQUIRE(prog.s_first >= 0 && prog.s_first < prog.S, "pchain: empty step range [%d, %d)", prog.s_first, prog.S);
  Hdr h{prog.ndesc, prog.s_first, prog.S, prog.B, prog.xcd, prog.prof_wg, prog.lds_products, prog.ctl, prog.prof};
  const int nw = prog.rt_group > 1 ? 8 : pchain_waves();  // (the row-group kernel is built for 8 waves: 512 threads finish row-tile pairs)
  const size_t lds_fixed = sizeof(int) * kDescWords * pchain::kMaxDesc + 32 * sizeof(unsigned long long);
  const size_t lds = lds_fixed + sizeof(float) * 2 * (size_t)prog.lds_products * nw * 256;
  // the dynamic-LDS limit of the kernels is raised once per process and device (the call is far from free), and the launch's
  // residency — every workgroup of the grid must be on a CU at the same time — is checked against the occupancy the runtime
  // computes for this kernel, block size and LDS size (cached per size)
  static std::mutex attr_mu;
  static int attr_dev[6] = {-1, -1, -1, -1, -1, -1};
  static size_t occ_lds[6] = {0, 0, 0, 0, 0, 0};
  static int occ_blocks[6] = {0, 0, 0, 0, 0, 0};
  const size_t lds_max = lds_fixed + sizeof(float) * 2 * (prog.rt_group > 1 ? 8 : 4) * (size_t)nw * 256;  // (row groups reduce two row tiles per barrier)
  BLVM_REQUIRE(lds <= lds_max, "pchain: %d products per tile exceed the reduction scratch", prog.lds_products);
  auto go = [&](auto kernel, int slot, int threads) -> int {
    {
      std::lock_guard<std::mutex> lock(attr_mu);
      if (attr_dev[slot] != dev) {
        BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_dev[slot] = dev;
        occ_lds[slot] = 0;
      }
      if (occ_lds[slot] != lds) {
        int per_cu = 0;
        BLVM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
        occ_blocks[slot] = per_cu;
        occ_lds[slot] = lds;
      }
      BLVM_REQUIRE((long)occ_blocks[slot] * cus >= grid,
                   "pchain: %d workgroups of %d threads + %zu B LDS are not co-resident (%d per CU x %d CUs)", grid, threads, lds, occ_blocks[slot], cus);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, (const int*)tab, h);
    return BLVM_OK;
  };
  BLVM_REQUIRE(prog.rt_group == 1 || prog.rt_group == 2 || prog.rt_group == 4, "pchain: row groups of %d row tiles are not built", prog.rt_group);
  if (prog.rt_group > 1) {
    for (int i = 0; i < prog.ndesc; ++i) {
      const int k = prog.d[i].kind;
      BLVM_REQUIRE(k == pchain::K_LIN || k == pchain::K_LINSEQ || k == pchain::K_HEAD || k == pchain::K_GRU || k == pchain::K_DZ || k == pchain::K_GRUB,
                   "pchain: tile kind %d has no row-group form", k);
      BLVM_REQUIRE(k != pchain::K_DZ || prog.d[i].p[2] == nullptr, "pchain: the row-group dz tile is the single-product form");
    }
    BLVM_REQUIRE(!prog.bf16, "pchain: row groups multiply fp32 operands only");
    const int rcg = prog.rt_group == 4 ? go(&pchain_rt_kernel<8, false, 4>, 4, 512) : go(&pchain_rt_kernel<8, false, 2>, 5, 512);
    if (rcg) return rcg;
    BLVM_CHECK_LAUNCH("pchain_launch (row groups)");
    return BLVM_OK;
  }
  int rc;
  if (nw == 16) rc = prog.bf16 ? go(&pchain_kernel<16, true>, 3, 1024) : go(&pchain_kernel<16, false>, 1, 1024);
  else rc = prog.bf16 ? go(&pchain_kernel<8, true>, 2, 512) : go(&pchain_kernel<8, false>, 0, 512);
  if (rc) return rc;
  BLVM_CHECK_LAUNCH("pchain_launch");
  return BLVM_OK;
}

}  // namespace blvm

// Diagnostics / unit test of the engine: a chain of L links x_{s+1} = relu(x_s W^T + b), [B,N] x [N,N], as a one-descriptor program.
// x16: (L+1) T16 slabs [ceil(B/16)*16, N] — slab 0 must hold x_0 (pchain_rows_to_t16 / blvm_pchain_rows_to_t16), slabs 1..L are
// sentinel-filled here; xs: L row-major slabs [B,N] (outputs).  W in the T16 operand layout.
extern "C" int blvm_pchain_chain_probe(const float* W16, const float* bias, float* x16, float* xs, int B, int N, int L, int nwg, void* stream_) {
  using namespace blvm;
  using namespace blvm::pchain;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(W16 && bias && x16 && xs && B > 0 && N > 0 && N % 16 == 0 && L > 0, "pchain_chain_probe: bad arguments");
  const int rt = (B + 15) / 16;
  const long x = (long)rt * 16 * N, sN = (long)B * N;
  Builder bld;
  bld.p.S = L; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 1;
  bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = 1;
  static const int run = [] { const char* e = getenv("BLVM_PCHAIN_PROBE_RUN"); return e ? atoi(e) : 1; }();  // links per K_LINSEQ visit
  const int nw = nwg > 0 ? nwg : range_for((N / 16) * rt, device_cus() & ~7);
  if (run >= 2 && run <= 4 && L % run == 0) {  // the same chain as runs of `run` links in one descriptor visit
    bld.p.S = L / run;
    SeqLink lk[4];
    for (int i = 0; i < run; ++i) lk[i] = SeqLink{W16, bias, xs + (long)i * sN, run * sN, N, x16 + (long)(i + 1) * x};
    add_linseq(bld, N / 16, 0, nw, N, true, false, 0, L / run, x16, run * x, run, lk, 0, run * x, N / 16, 0.f, 0);
  } else {
    Desc& d = bld.add(K_LIN, N / 16, 0, nw, N, DF_RELU, 0, L);
    bld.ptr(d, 0, x16, x); bld.ptr(d, 1, W16); bld.ptr(d, 2, bias); bld.ptr(d, 5, xs, sN); bld.ptr(d, 6, x16 + x, x);
    d.ld[3] = N; d.n16[0] = N / 16; d.f[0] = 0.f;
  }
  int rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
  if (rc) return rc;
  BLVM_HIP(pchain_fill_sentinel(x16 + x, sizeof(float) * (size_t)x * L, s));
  return pchain_launch(bld.p, s);
}

extern "C" int blvm_pchain_rows_to_t16(const float* src, int ld, int B, int K, float* dst, void* stream) {
  return blvm::pchain_rows_to_t16(src, ld, B, K, dst, static_cast<hipStream_t>(stream));
}
