// pchain.hip — the persistent-chain engine: ONE kernel that executes a host-built program of link descriptors (pchain.h) for a
// whole recurrent sequence.  Replaces, per model, thousands of dependent launches (VRNN [64,16000]: 4 500 per train step) by one
// forward and one backward launch.  Pointer roles of a descriptor by tile kind:
//
//   K_LIN   p: 0 A (T16 copy, polled | row-major when DF_A_PLAIN, ld[0])  1 W (T16)  2 bias  3 add (ld[1])  4 gate (ld[2])
//              5 out row-major (ld[3])  6 out T16 (n16[0])            f: 0 slope
//   K_HEAD  p: 0 P16  1 Q16  2 Wp  3 bp  4 Wq  5 bq  6 eps  7 mu_p  8 sd_p  9 mu_q  10 sd_q  11 raw_p  12 raw_q  13 muq_raw
//              14 z row-major (ld[3])  15 z T16 (n16[0])               i: 0 Z  1 residual      f: 0 beta  1 1/beta  2 sd_eps
//   K_GRU   p: 0 X16  1 Wih (T16)  2 xg  3 gh (polled words)  4 h_prev (polled words, ld[0])  5 h_new row-major (ld[3])
//              6 h_new T16 (n16[0])  7 rg  8 ug  9 ng                   i: 0 R
//   K_DZ    p: 0 D16  1 WT  2 D2_16  3 WT2  4 dz_add (ld[1])  5 mu_q  6 sd_q  7 mu_p  8 sd_p  9 eps  10 raw_q  11 raw_p  12 muq_raw
//              13 x_sl (int32)  14 c_raw  15 c_fn  16 dqh row-major  17 dqh T16  18 dph row-major  19 dph T16   (ld[3] = 2Z, n16[0])
//              i: 0 Z  1 residual  2 stride  3 t at s = 0 (t = i[3] - s)   f: 0 fn_floor  1 beta  2 sd_eps
//   K_GRUB  p: 0 D0_16  1 D1_16  2 W0  3 W1  4 g_in (polled words)  5 rg  6 ug  7 ng  8 gh  9 h_prev (ld[0])  10 dd (ld[0])
//              11 dgi row-major  12 dgi T16  13 dgh row-major  14 dgh T16  (ld[3] = 3R, n16[0])  15 ga  16 g_out
//              i: 0 R  1 first step with the products  2 first step WITHOUT gates
#include <algorithm>
#include <mutex>

#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {
using namespace pchain;

// A descriptor as the kernel reads it: strides resolved per pointer, in device memory written by pchain_resolve_kernel right before
// the launch and never written again, so that the persistent kernel fetches a tile's operands with a handful of independent SCALAR
// loads (s_load_dwordx8/x16 through the scalar cache, results uniform in SGPRs).  Read straight from the kernel argument it was a
// chain of dependent scalar loads (descriptor -> stride index -> stride); from an LDS copy, ~15 vector LDS reads + readfirstlanes.
struct RDesc {
  int kind, ct, wg0, nwg, flags, K, s_begin, s_end;
  int ld[4];
  int n16[2];
  int i[4];
  float f[4];
  int pad[2];
  const float* p[kMaxPtr];
  long st[kMaxPtr];
};
struct Hdr {
  int ndesc, S, B, xcd, prof_wg, lds_products;
  Ctl ctl;
  unsigned long long* prof;
};

__global__ void pchain_resolve_kernel(Program a, RDesc* __restrict__ out) {
  for (int e = threadIdx.x; e < a.ndesc * kMaxPtr; e += blockDim.x) {
    const int i = e / kMaxPtr, k = e % kMaxPtr;
    out[i].p[k] = a.d[i].p[k];
    out[i].st[k] = a.stride[a.d[i].sidx[k]];
  }
  for (int i = threadIdx.x; i < a.ndesc; i += blockDim.x) {
    const Desc& d = a.d[i];
    RDesc& l = out[i];
    l.kind = d.kind; l.ct = d.ct; l.wg0 = d.wg0; l.nwg = d.nwg; l.flags = d.flags; l.K = d.K; l.s_begin = d.s_begin; l.s_end = d.s_end;
    for (int k = 0; k < 4; ++k) { l.ld[k] = d.ld[k]; l.i[k] = d.i[k]; l.f[k] = d.f[k]; }
    l.n16[0] = d.n16[0]; l.n16[1] = d.n16[1];
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void pchain_kernel(const RDesc* __restrict__ L, Hdr a) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];  // [profile | 2 x (lds_products x NW x 256) floats]
  unsigned long long* const pacc = reinterpret_cast<unsigned long long*>(lds_all);
  float* const red0 = reinterpret_cast<float*>(lds_all + 16 * sizeof(unsigned long long));
  float* const red1 = red0 + a.lds_products * NW * 256;
  const int w = blockIdx.x, B = a.B, rt = (B + 15) / 16;
  const bool xcd = a.xcd != 0;
  if (threadIdx.x < 16) pacc[threadIdx.x] = 0ull;
  unsigned mine = 0;  // the descriptors this workgroup has tiles of
  for (int i = 0; i < a.ndesc; ++i)
    if (TileIter(w, L[i].wg0, L[i].nwg, rt, L[i].ct, xcd).valid()) mine |= 1u << i;
  __syncthreads();
  int par = 0;
  auto red = [&]() { par ^= 1; return par ? red0 : red1; };
  Poll pl{a.ctl, 0u, false, 1};
  const bool profiled = a.prof != nullptr && (w == 0 || w == a.prof_wg) && threadIdx.x == 0;
  unsigned long long tprev = profiled ? wall_clock64() : 0ull;
  for (int s = 0; s < a.S; ++s) {
    for (unsigned m = mine; m != 0; m &= m - 1) {
      const int i = __builtin_ctz(m);
      const RDesc& d = L[i];
      if (s < d.s_begin || s >= d.s_end) continue;
      const int kind = d.kind, flags = d.flags, K = d.K;
      TileIter it(w, d.wg0, d.nwg, rt, d.ct, xcd);
      pl.nap = (flags & DF_GENTLE) ? 16 : 1;
      pl.code = ((unsigned)s << 4) | (unsigned)i;
      auto P = [&](int k) -> const float* { return d.p[k] + (long)s * d.st[k]; };
      auto M = [&](int k) -> float* { return const_cast<float*>(d.p[k] + (long)s * d.st[k]); };
      auto Pn = [&](int k) -> const float* { const float* q = d.p[k]; return q ? q + (long)s * d.st[k] : nullptr; };
      auto Mn = [&](int k) -> float* { return const_cast<float*>(Pn(k)); };
      switch (kind) {
        case K_LIN: {
          const bool a_polled = !(flags & DF_A_PLAIN);
          const float *A = P(0), *W = d.p[1], *bias = d.p[2], *add = Pn(3), *gate = Pn(4);
          const Out o{Mn(5), (d.ld[3]), (flags & DF_RM_SC1) != 0, Mn(6), (d.n16[0])};
          const int lda = (d.ld[0]), ldadd = (d.ld[1]), ldgate = (d.ld[2]);
          const float slope = d.f[0];
          for (; it.valid(); it.next()) {
            if ((flags & DF_CANARY) && a_polled) canary_wait(A, it.r0(), K, pl);
            tile_lin<NW>(A, lda, a_polled, W, K, bias, add, ldadd, (flags & DF_ADD_POLLED) != 0, gate, ldgate, (flags & DF_RELU) != 0, slope, o, it.r0(),
                         it.c() * 16, B, red(), pl);
          }
        } break;
        case K_HEAD: {
          const HeadOut o{M(7), M(8), M(9), M(10), M(11), M(12), Mn(13), Out{M(14), (d.ld[3]), false, M(15), (d.n16[0])}};
          const int Z = (d.i[0]), residual = (d.i[1]);
          for (; it.valid(); it.next())
            tile_head<NW>(P(0), P(1), true, d.p[2], d.p[3], d.p[4], d.p[5], P(6), o, K, Z, residual, d.f[0], d.f[1], d.f[2], it.r0(), it.c() * 16, B, red(), pl);
        } break;
        case K_GRU: {
          const Out o{M(5), (d.ld[3]), true, M(6), (d.n16[0])};
          const int R = (d.i[0]), ldh = (d.ld[0]);
          for (; it.valid(); it.next())
            tile_gru<NW>(P(0), 0, true, d.p[1], K, Pn(2), P(3), P(4), ldh, R, o, M(7), M(8), M(9), it.r0(), it.c() * 16, B, red(), pl);
        } break;
        case K_DZ: {
          DzIn z;
          z.mu_q = P(5); z.sd_q = P(6); z.mu_p = P(7); z.sd_p = P(8); z.eps = P(9); z.raw_q = P(10); z.raw_p = P(11);
          z.muq_raw = Pn(12);
          z.x_sl = reinterpret_cast<const int32_t*>(d.p[13]); z.c_raw = d.p[14]; z.c_fn = d.p[15];
          z.t = (d.i[3]) - s; z.stride = (d.i[2]); z.residual = (d.i[1]);
          z.fn_floor = d.f[0]; z.beta = d.f[1]; z.sd_eps = d.f[2];
          const int ldo = (d.ld[3]), n16 = (d.n16[0]), Z = (d.i[0]);
          const Out oq{M(16), ldo, false, M(17), n16}, op{M(18), ldo, false, M(19), n16};
          for (; it.valid(); it.next())
            tile_dz<NW>(P(0), d.p[1], nullptr, nullptr, true, Pn(4), (d.ld[1]), (flags & DF_ADD_POLLED) != 0, z, oq, op, K, Z, it.r0(), it.c() * 16, B,
                        red(), pl);
        } break;
        case K_GRUB: {
          GrubIn g;
          g.D0 = P(0); g.D1 = P(1); g.W0 = d.p[2]; g.W1 = d.p[3]; g.g_in = P(4);
          g.rg = P(5); g.ug = P(6); g.ng = P(7); g.gh = P(8); g.hprev = P(9); g.dd = P(10); g.ldh = (d.ld[0]);
          const int ldo = (d.ld[3]), n16 = (d.n16[0]);
          g.dgi = Out{M(11), ldo, false, M(12), n16};
          g.dgh = Out{M(13), ldo, false, M(14), n16};
          g.ga = M(15); g.g_out = const_cast<float*>(d.p[16]);
          g.has_gemm = s >= (d.i[1]); g.has_gates = s < (d.i[2]);
          const int R = (d.i[0]);
          for (; it.valid(); it.next()) tile_grub<NW>(g, K, R, it.r0(), it.c() * 16, B, red(), pl);
        } break;
        default: break;
      }
      if (profiled) {
        const unsigned long long now = wall_clock64();
        pacc[i] += now - tprev;
        tprev = now;
      }
    }
  }
  if (profiled) {
    for (int i = 0; i < a.ndesc; ++i) a.prof[(w == 0 ? 0 : 16) + i] += pacc[i];
#ifdef PCHAIN_TPROF
    if (w == 0) for (int l = 0; l < 8; ++l) a.prof[32 + l] += pl.tp[l];
#endif
  }
}

}  // namespace

int pchain_launch(const pchain::Program& prog, hipStream_t stream) {
  BLVM_REQUIRE(prog.ndesc > 0 && prog.ndesc <= pchain::kMaxDesc && prog.S > 0 && prog.B > 0, "pchain: bad program (%d descriptors, %d steps)", prog.ndesc, prog.S);
  int grid = 0;
  for (int i = 0; i < prog.ndesc; ++i) {
    const pchain::Desc& d = prog.d[i];
    BLVM_REQUIRE(d.nwg > 0 && d.wg0 >= 0 && d.ct > 0 && d.K > 0 && d.K % 16 == 0, "pchain: bad descriptor %d", i);
    BLVM_REQUIRE(!prog.xcd || d.nwg % 8 == 0, "pchain: XCD-aware placement needs ranges of 8 k workgroups (descriptor %d has %d)", i, d.nwg);
    grid = std::max(grid, d.wg0 + d.nwg);
  }
  int dev = 0, cus = 0;
  BLVM_HIP(hipGetDevice(&dev));
  BLVM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  BLVM_REQUIRE(grid <= cus, "pchain: the program names %d workgroups, the device has %d CUs (every workgroup must be resident)", grid, cus);
  // the resolved table: a slot of a small ring in library-owned device memory (launches on one stream are ordered; the ring keeps
  // launches that overlap on different streams apart)
  RDesc* tab = nullptr;
  {
    static std::mutex mu;
    static int tab_dev = -1;
    static RDesc* ring = nullptr;
    static unsigned next = 0;
    constexpr unsigned kSlots = 8;
    std::lock_guard<std::mutex> lock(mu);
    if (tab_dev != dev) {
      BLVM_HIP(hipMalloc(reinterpret_cast<void**>(&ring), sizeof(RDesc) * pchain::kMaxDesc * kSlots));
      tab_dev = dev;
    }
    tab = ring + (size_t)(next++ % kSlots) * pchain::kMaxDesc;
  }
  hipLaunchKernelGGL(pchain_resolve_kernel, dim3(1), dim3(256), 0, stream, prog, tab);
  Hdr h{prog.ndesc, prog.S, prog.B, prog.xcd, prog.prof_wg, prog.lds_products, prog.ctl, prog.prof};
  const int nw = pchain_waves();
  const size_t lds = 16 * sizeof(unsigned long long) + sizeof(float) * 2 * (size_t)prog.lds_products * nw * 256;
  if (nw == 16) {
    BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pchain_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pchain_kernel<16>), dim3(grid), dim3(1024), lds, stream, (const RDesc*)tab, h);
  } else {
    BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pchain_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pchain_kernel<8>), dim3(grid), dim3(512), lds, stream, (const RDesc*)tab, h);
  }
  BLVM_CHECK_LAUNCH("pchain_launch");
  return BLVM_OK;
}

}  // namespace blvm
