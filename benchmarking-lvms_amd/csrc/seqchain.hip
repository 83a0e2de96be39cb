// seqchain.hip — register-resident persistent kernels for GRU / LSTM sequences (see seqchain.h).
#include "seqchain.h"

namespace blvm {
namespace {
using namespace pchain;
constexpr int NW = 8;

// acc[g] += A[r0+i][k] W[g][c0[g]+j][k] over this wave's NCH k-chunks (chunk numbers wave, wave + NW, ...): A polled from a T16 slab
// (sentinel protocol of pchain.h), W from the registers `w` (loaded once by load_w).  `mid` runs behind the first poll's loads.
template <bool BF, int G, int NCH>
__device__ __forceinline__ void load_w(typename WFrag<BF>::type (&w)[G][NCH], const float* W, const int (&c0)[G], int K) {
  typedef typename WFrag<BF>::type wfrag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ES = BF ? 2 : 4;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const char* base = reinterpret_cast<const char*>(W) + (size_t)ES * ((size_t)c0[g] * K + 4 * lane);
#pragma unroll
    for (int u = 0; u < NCH; ++u) w[g][u] = *reinterpret_cast<const wfrag*>(base + (size_t)ES * 256 * (size_t)(wave + u * NW));
  }
}
// SHARED (wide operands, K >= 1024): every tile of a row tile reads the same [16, K] slab, and the workgroups of one XCD all belong
// to the same row tile (blockIdx % 8 fixes blockIdx % rt for rt = 1, 2, 4, 8) — so once one wave has seen the slab's canary words
// arrive (sc1 polls of word 0 of every 1 KB block), the fragments are read with ORDINARY loads: the first workgroup of the XCD
// brings a line into the XCD's L2, the others hit it, and the fabric carries the slab once per XCD instead of once per tile.  Every
// word is still validated: a fragment that holds a sentinel (a line cached before its last store landed) is re-read with sc1 loads,
// which bypass the stale line.
template <bool BF, int G, int NCH, bool SHARED, class Mid>
__device__ __forceinline__ void product(const float* A16, int r0, int nrows, int K, const typename WFrag<BF>::type (&w)[G][NCH], f32x4 (&acc)[G], Poll& pl,
                                        Mid mid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool aok = (r0 + (lane & 15)) < nrows;
  const rsrc_t ar = make_rsrc(A16);
  const unsigned aoff = 4u * ((unsigned)(r0 >> 4) * 16u * (unsigned)K + 4u * (unsigned)lane) + 1024u * (unsigned)wave;
  f32x4 a[NCH];
  unsigned spins = 0;
  bool mid_pending = true;
  if constexpr (SHARED) {
    canary_wait(A16, r0, K, pl);  // (one wave polls; ends in a workgroup barrier)
    const float* ap = A16 + (size_t)(r0 >> 4) * 16 * K + 4 * lane + 256 * wave;
#pragma unroll
    for (int u = 0; u < NCH; ++u) a[u] = *reinterpret_cast<const f32x4*>(ap + 256 * (size_t)(u * NW));
    mid();
    mid_pending = false;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < NCH; ++u) bad |= any_sentinel(a[u]);
    if (!__any(bad && aok)) goto multiply;
  }
  for (;;) {
#pragma unroll
    for (int u = 0; u < NCH; ++u) a[u] = ld_sc1_x4(ar, aoff + 1024u * (unsigned)(u * NW));
    if (mid_pending) { mid(); mid_pending = false; }
    bool bad = false;
#pragma unroll
    for (int u = 0; u < NCH; ++u) bad |= any_sentinel(a[u]);
    if (!__any(bad && aok) || pl.dead) break;
    if (spin_tick(spins, pl.ctl, pl.code, pl.dead)) break;
    pl.sleep();
  }
multiply:
  if constexpr (BF) {
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const u32x2 q = {aok ? pk_bf16(a[u][0], a[u][1]) : 0u, aok ? pk_bf16(a[u][2], a[u][3]) : 0u};
      const s16x4 ab = __builtin_bit_cast(s16x4, q);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, __builtin_bit_cast(s16x4, w[g][u]), acc[g], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < NCH; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(aok ? a[u][e] : 0.f, w[g][u][e], acc[g], 0, 0, 0);
  }
}

struct TileAt { int r0, c0; };
__device__ __forceinline__ TileAt my_tile(int B) {
  const int rt = (B + 15) / 16;
  return TileAt{(int)(blockIdx.x % rt) * 16, (int)(blockIdx.x / rt) * 16};
}

template <bool BF, int NCH>
__global__ __launch_bounds__(NW * 64, 1) void gru_fwd_kernel(SeqGruFwd a) {
  __shared__ float red[2][3 * NW * 256];
  const TileAt tl = my_tile(a.B);
  const int R = a.R, B = a.B, r0 = tl.r0, c0 = tl.c0;
  const int t = threadIdx.x & 255, row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col, sR = (size_t)B * R, xR = (size_t)((B + 15) / 16) * 16 * R;
  typename WFrag<BF>::type w[3][NCH];
  {
    const int cs[3] = {c0, R + c0, 2 * R + c0};
    load_w<BF, 3, NCH>(w, a.Whh, cs, R);
  }
  const float b0 = a.bhh[col], b1 = a.bhh[R + col], b2 = a.bhh[2 * R + col];
  Poll pl{a.ctl, 0u, false, 1};
  const Out hn{nullptr, R, false, nullptr, R / 16};
  for (int j = 0; j < a.T; ++j) {
    pl.code = ((unsigned)j << 4) | 1u;
    int idx = 0;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, hp = 0.f;
    auto prefetch = [&]() {
      idx = seq_time_index(j, a.reverse, a.lens, rowc);
      const size_t ox = ((size_t)idx * B + rowc) * 3 * R + col;
      x0 = a.xg[ox]; x1 = a.xg[ox + R]; x2 = a.xg[ox + 2 * R];
      hp = a.Hs[(size_t)j * sR + o];
    };
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    product<BF, 3, NCH, false>(a.H16 + (size_t)j * xR, r0, B, R, w, acc, pl, prefetch);
    float v[3];
    reduce_tiles<3, NW>(acc, red[j & 1], v);
    if (!own) continue;
    const float hnn = v[2] + b2;
    const float r = sigmoidf_(x0 + v[0] + b0);
    const float u = sigmoidf_(x1 + v[1] + b1);
    const float n = tanhf(x2 + r * hnn);
    const float h2 = (1.f - u) * n + u * hp;
    Out oo = hn;
    oo.rm = a.Hs + (size_t)(j + 1) * sR; oo.x16 = const_cast<float*>(a.H16) + (size_t)(j + 1) * xR;
    put(oo, r0, c0, row, col, h2);
    a.out[(size_t)idx * a.out_ts + (size_t)row * a.out_ld + col] = h2;
    const size_t os = (size_t)j * sR + o;
    a.rg[os] = r; a.ug[os] = u; a.ng[os] = n; a.ghn[os] = hnn;
  }
}

template <bool BF, int NCH>
__global__ __launch_bounds__(NW * 64, 1) void gru_bwd_kernel(SeqGruBwd a) {
  __shared__ float red[2][NW * 256];
  const TileAt tl = my_tile(a.B);
  const int R = a.R, B = a.B, T = a.T, r0 = tl.r0, c0 = tl.c0;
  const int t = threadIdx.x & 255, row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * R + col, sR = (size_t)B * R, x3R = (size_t)((B + 15) / 16) * 16 * 3 * R;
  typename WFrag<BF>::type w[1][NCH];
  {
    const int cs[1] = {c0};
    load_w<BF, 1, NCH>(w, a.WhhT, cs, 3 * R);
  }
  Poll pl{a.ctl, 0u, false, 1};
  float g = own ? a.G[o] : 0.f;  // the running gradient through the u-gate path: this thread's for the whole sequence
  for (int s = 0; s < a.steps; ++s) {
    const int j = T - 1 - s;
    const bool has_gemm = s >= 1, has_gates = s < T;
    pl.code = ((unsigned)s << 4) | 2u;
    float dout = 0.f, r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f;
    int idx = 0;
    auto prefetch = [&]() {
      if (has_gates) {  // uniform
        idx = seq_time_index(j, a.reverse, a.lens, rowc);
        dout = a.dout[(size_t)idx * a.out_ts + (size_t)rowc * a.out_ld + col];
        const size_t os = (size_t)j * sR + o;
        r = a.rg[os]; u = a.ug[os]; n = a.ng[os]; hn = a.ghn[os]; hp = a.Hs[os];
      }
    };
    float v[1] = {0.f};
    if (has_gemm) {  // uniform
      f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      product<BF, 1, NCH, (NCH >= 12)>(a.DGH16 + (size_t)(s - 1) * x3R, r0, B, 3 * R, w, acc, pl, prefetch);
      reduce_tiles<1, NW>(acc, red[s & 1], v);
    } else {
      prefetch();
    }
    if (!own) continue;
    g += dout + v[0];
    if (!has_gates) { a.dh0[o] = g; continue; }
    const float dn_pre = g * (1.f - u) * (1.f - n * n);
    const float du_pre = g * (hp - n) * u * (1.f - u);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    const size_t oi = ((size_t)idx * B + row) * 3 * R + col;
    a.DGI[oi] = dr_pre; a.DGI[oi + R] = du_pre; a.DGI[oi + 2 * R] = dn_pre;
    const Out od{a.DGH + (size_t)j * 3 * sR, 3 * R, false, const_cast<float*>(a.DGH16) + (size_t)s * x3R, 3 * R / 16};
    put(od, r0, c0, row, col, dr_pre);
    put(od, r0, R + c0, row, R + col, du_pre);
    put(od, r0, 2 * R + c0, row, 2 * R + col, dn_pre * r);
    g *= u;
  }
  if (own) a.G[o] = g;
}

template <bool BF, int NCH>
__global__ __launch_bounds__(NW * 64, 1) void lstm_fwd_kernel(SeqLstmFwd a) {
  __shared__ float red[2][4 * NW * 256];
  const TileAt tl = my_tile(a.B);
  const int H = a.H, B = a.B, r0 = tl.r0, c0 = tl.c0;
  const int tt = threadIdx.x & 255, row = r0 + (tt >> 4), col = c0 + (tt & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col, sH = (size_t)B * H, xH = (size_t)((B + 15) / 16) * 16 * H;
  typename WFrag<BF>::type w[4][NCH];
  {
    const int cs[4] = {c0, H + c0, 2 * H + c0, 3 * H + c0};
    load_w<BF, 4, NCH>(w, a.Whh, cs, H);
  }
  const float b0 = a.bhh[col], b1 = a.bhh[H + col], b2 = a.bhh[2 * H + col], b3 = a.bhh[3 * H + col];
  const int len = a.lens != nullptr ? a.lens[rowc] : a.T;
  float cp = a.Cs[o];  // the cell state of this element lives in a register for the whole sequence (and is saved per step)
  Poll pl{a.ctl, 0u, false, 1};
  for (int t = 0; t < a.T; ++t) {
    pl.code = ((unsigned)t << 4) | 3u;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f, hp = 0.f;
    auto prefetch = [&]() {
      const float* xg = a.xg + (size_t)t * 4 * sH;
      x0 = xg[o4] + b0; x1 = xg[o4 + H] + b1; x2 = xg[o4 + 2 * H] + b2; x3 = xg[o4 + 3 * H] + b3;
      hp = a.Hs[(size_t)t * sH + o];
    };
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    product<BF, 4, NCH, false>(a.H16 + (size_t)t * xH, r0, B, H, w, acc, pl, prefetch);
    float v[4];
    reduce_tiles<4, NW>(acc, red[t & 1], v);
    if (!own) continue;
    const bool live = t < len;
    const float i = sigmoidf_(v[0] + x0), f = sigmoidf_(v[1] + x1), g = tanhf(v[2] + x2), og = sigmoidf_(v[3] + x3);
    const float c2 = f * cp + i * g;
    const float h2 = og * tanhf(c2);
    cp = live ? c2 : cp;
    a.Cs[(size_t)(t + 1) * sH + o] = cp;
    const Out oo{a.Hs + (size_t)(t + 1) * sH, H, false, const_cast<float*>(a.H16) + (size_t)(t + 1) * xH, H / 16};
    put(oo, r0, c0, row, col, live ? h2 : hp);
    a.out[(size_t)t * sH + o] = live ? h2 : 0.f;
    float* gt = a.gates + (size_t)t * 4 * sH;
    gt[o4] = live ? i : 0.f; gt[o4 + H] = live ? f : 0.f; gt[o4 + 2 * H] = live ? g : 0.f; gt[o4 + 3 * H] = live ? og : 0.f;
  }
}

template <bool BF, int NCH>
__global__ __launch_bounds__(NW * 64, 1) void lstm_bwd_kernel(SeqLstmBwd a) {
  __shared__ float red[2][NW * 256];
  const TileAt tl = my_tile(a.B);
  const int H = a.H, B = a.B, T = a.T, r0 = tl.r0, c0 = tl.c0;
  const int tt = threadIdx.x & 255, row = r0 + (tt >> 4), col = c0 + (tt & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o4 = (size_t)rowc * 4 * H + col, sH = (size_t)B * H, x4H = (size_t)((B + 15) / 16) * 16 * 4 * H;
  typename WFrag<BF>::type w[1][NCH];
  {
    const int cs[1] = {c0};
    load_w<BF, 1, NCH>(w, a.WhhT, cs, 4 * H);
  }
  Poll pl{a.ctl, 0u, false, 1};
  float dc = own ? a.DC[o] : 0.f;  // running gradient wrt the cell state
  for (int s = 0; s < a.steps; ++s) {
    const int t = T - 1 - s;
    const bool has_gemm = s >= 1, has_gates = s < T;
    pl.code = ((unsigned)s << 4) | 4u;
    float dh = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cs = 0.f, cs1 = 0.f;
    auto prefetch = [&]() {
      if (has_gates) {  // uniform
        dh = a.dout[(size_t)t * sH + o];
        const float* gt = a.gates + (size_t)t * 4 * sH;
        ig = gt[o4]; fg = gt[o4 + H]; gg = gt[o4 + 2 * H]; og = gt[o4 + 3 * H];
        cs = a.Cs[(size_t)t * sH + o]; cs1 = a.Cs[(size_t)(t + 1) * sH + o];
      }
    };
    float v[1] = {0.f};
    if (has_gemm) {  // uniform
      f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      product<BF, 1, NCH, (NCH >= 12)>(a.DG16 + (size_t)(s - 1) * x4H, r0, B, 4 * H, w, acc, pl, prefetch);
      reduce_tiles<1, NW>(acc, red[s & 1], v);
    } else {
      prefetch();
    }
    if (!own) continue;
    dh += v[0];
    if (!has_gates) { a.dh0[o] = dh; continue; }
    const float tc = tanhf(cs1);
    const float d_o = dh * tc;
    const float dct = dc + dh * og * (1.f - tc * tc);
    const Out od{a.DG + (size_t)t * 4 * sH, 4 * H, false, const_cast<float*>(a.DG16) + (size_t)s * x4H, 4 * H / 16};
    put(od, r0, c0, row, col, dct * gg * ig * (1.f - ig));
    put(od, r0, H + c0, row, H + col, dct * cs * fg * (1.f - fg));
    put(od, r0, 2 * H + c0, row, 2 * H + col, dct * ig * (1.f - gg * gg));
    put(od, r0, 3 * H + c0, row, 3 * H + col, d_o * og * (1.f - og));
    dc = dct * fg;
  }
  if (own) a.DC[o] = dc;
}

inline bool nch_ok(int nch) { return nch == 1 || nch == 2 || nch == 3 || nch == 4 || nch == 6 || nch == 8 || nch == 12; }

#define SEQ_LAUNCH(kernel, nch, bf, grid, s, args)                                                                                  \
  do {                                                                                                                              \
    switch (nch) {                                                                                                                  \
      case 1: if (bf) hipLaunchKernelGGL((kernel<true, 1>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 1>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 2: if (bf) hipLaunchKernelGGL((kernel<true, 2>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 2>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 3: if (bf) hipLaunchKernelGGL((kernel<true, 3>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 3>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 4: if (bf) hipLaunchKernelGGL((kernel<true, 4>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 4>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 6: if (bf) hipLaunchKernelGGL((kernel<true, 6>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 6>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 8: if (bf) hipLaunchKernelGGL((kernel<true, 8>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 8>), grid, dim3(NW * 64), 0, s, args); break;   \
      case 12: if (bf) hipLaunchKernelGGL((kernel<true, 12>), grid, dim3(NW * 64), 0, s, args); else hipLaunchKernelGGL((kernel<false, 12>), grid, dim3(NW * 64), 0, s, args); break; \
      default: set_error("seqchain: %d k-chunks per wave are not instantiated", nch); return BLVM_ENOSUP;                           \
    }                                                                                                                               \
  } while (0)

}  // namespace

// env BLVM_SEQ_REGS: bit 0 forward, bit 1 backward sequences on these kernels (default: see seq_regs_mask)
int seq_regs_mask() {
  static const int v = [] { const char* e = getenv("BLVM_SEQ_REGS"); return e ? atoi(e) : 3; }();
  return v;
}
bool seq_regs_applies(int K_fwd, int K_bwd, int hidden, int B, int gates) {
  if (K_fwd % (NW * 16) != 0 || K_bwd % (NW * 16) != 0 || hidden % 16 != 0) return false;
  if (!nch_ok(K_fwd / (NW * 16)) || !nch_ok(K_bwd / (NW * 16))) return false;
  if (gates * (K_fwd / (NW * 16)) > 24) return false;  // the forward slice (gates x k-chunks fragments per lane) must fit the registers
  return (hidden / 16) * ((B + 15) / 16) <= device_cus();
}

int seq_gru_fwd(const SeqGruFwd& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.R / 16) * ((a.B + 15) / 16)));
  SEQ_LAUNCH(gru_fwd_kernel, a.R / (NW * 16), a.bf16 != 0, grid, s, a);
  BLVM_CHECK_LAUNCH("seq_gru_fwd");
  return BLVM_OK;
}
int seq_gru_bwd(const SeqGruBwd& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.R / 16) * ((a.B + 15) / 16)));
  SEQ_LAUNCH(gru_bwd_kernel, 3 * a.R / (NW * 16), a.bf16 != 0, grid, s, a);
  BLVM_CHECK_LAUNCH("seq_gru_bwd");
  return BLVM_OK;
}
int seq_lstm_fwd(const SeqLstmFwd& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.H / 16) * ((a.B + 15) / 16)));
  SEQ_LAUNCH(lstm_fwd_kernel, a.H / (NW * 16), a.bf16 != 0, grid, s, a);
  BLVM_CHECK_LAUNCH("seq_lstm_fwd");
  return BLVM_OK;
}
int seq_lstm_bwd(const SeqLstmBwd& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.H / 16) * ((a.B + 15) / 16)));
  SEQ_LAUNCH(lstm_bwd_kernel, 4 * a.H / (NW * 16), a.bf16 != 0, grid, s, a);
  BLVM_CHECK_LAUNCH("seq_lstm_bwd");
  return BLVM_OK;
}

}  // namespace blvm
