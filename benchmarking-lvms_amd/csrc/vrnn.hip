// vrnn.hip — K1: the VRNN recurrent cell over a whole sequence, forward and BPTT, as hand-written gfx950 kernels.
//
// Replaces the scripted per-step loop of the reference (blvm/models/vrnn.py:305-308 calling VRNNCell.forward,
// vrnn.py:109-141: prior MLP + DiagonalGaussianDense, posterior MLP + DiagonalGaussianDense on cat[h,x], residual
// posterior, rsample, phi_z MLP, nn.GRUCell on cat[x,phi_z]) and the autograd backward of that loop.
//
// Structure (v1, "stage kernels"): every recurrent step is a chain of nine dependent small GEMMs with M = batch.
// Each link is ONE launch whose workgroups own a 16(batch) x 16(feature) output tile; the four waves of a
// workgroup split K and combine through LDS, so one launch spreads over (B/16) x (N/16) workgroups and a wave
// issues only K/64 v_mfma_f32_16x16x4_f32.  All element-wise work (bias, ReLU, softplus head, residual mean,
// reparameterisation, GRU gates, and in the backward pass the KL/free-nats gradient, activation derivatives and the
// GRU gate derivatives) is fused into the epilogue of the producing GEMM.  Everything that does not depend on the
// recurrent state is hoisted out of the loop into large MFMA GEMMs (gemm.hip): the x-halves of the posterior's
// first layer and of the GRU input projection before the loop; all weight gradients and d(enc) after it.
//
// Numerics: fp32 operands, fp32 MFMA accumulation (an exact fma chain), so a sequence is reproducible run-to-run.
#include <vector>
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {

#include "stages.h"

// ---------------------------------------------------------------------------------------------------------------
// F9: GRU input projection of phi + gates + state update
// ---------------------------------------------------------------------------------------------------------------
// arguments: decin_t [B,H+R] row block t of decin = [phi | h_prev]; Wih [3R,H] the phi columns of the GRU input weight, in T16;
// xg [B,3R] x-part of the input projection incl. b_ih; gh [B,3R] hidden projection incl. b_hh; decin_next = row block t+1 (h-part
// written); rg, ug, ng [B,R] saved gates

template <int NW>
__global__ __launch_bounds__(NW * 64) void gru_stage_kernel(const float* decin_t, const float* Wih, const float* xg,
                                                            const float* gh, int B, int H, int R, float* decin_next,
                                                            float* rg, float* ug, float* ng) {
  // scalar arguments: the operand pointers and sizes are the first 11 dwords, preloaded into SGPRs (stages.h lin1_stage_kernel)
  __shared__ float red[3 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6;
  const int ldd = H + R;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;  // clamped: unconditional prefetch
  const size_t o3 = (size_t)rowc * 3 * R + col;
  const float x0 = xg[o3], x1 = xg[o3 + R], x2 = xg[o3 + 2 * R];
  const float hr = gh[o3], hz = gh[o3 + R], hn = gh[o3 + 2 * R];
  const float hp = decin_t[(size_t)rowc * ldd + H + col];
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[3] = {decin_t, decin_t, decin_t};
    const float* const Ws[3] = {Wih, Wih, Wih};
    const int la[3] = {ldd, ldd, ldd}, lw[3] = {H, H, H}, cs[3] = {c0, R + c0, 2 * R + c0};
    wave_gemm16_multi<NW, 3, true>(As, la, r0, B, Ws, lw, cs, H, wave, acc);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (!own) return;
  const float r = sigmoidf_(v[0] + x0 + hr);
  const float u = sigmoidf_(v[1] + x1 + hz);
  const float n = tanhf(v[2] + x2 + r * hn);
  decin_next[(size_t)row * ldd + H + col] = (1.f - u) * n + u * hp;
  const size_t o = (size_t)row * R + col;
  rg[o] = r; ug[o] = u; ng[o] = n;
}

// ---------------------------------------------------------------------------------------------------------------
// B10 (+ gate derivatives of the PREVIOUS step): G <- G + DP0 W_p0 + DQ0 W_q0h, then GRU backward of step s = t-1
// ---------------------------------------------------------------------------------------------------------------
struct DhArgs {
  const float *DP0, *DQ0;   // [B,H]   (null when has_gemm == 0)
  const float *WpT, *WqT;   // [R,H] in T16
  float* G;                 // [B,R] running gradient wrt the recurrent state (in/out)
  // step s (the step whose OUTPUT state G refers to); has_gates == 0 for the very first state
  const float *rg, *ug, *ng, *gh;   // [B,R] x3, [B,3R]
  const float* decin_s;             // [B,H+R] (h-part = state entering step s)
  const float* ddecin_s;            // [B,H+R] decoder gradient wrt decin row s
  float *dgi, *dgh;                 // [B,3R]
  int B, H, R, has_gemm, has_gates;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void dh_stage_kernel(const float* DP0, const float* DQ0, const float* WpT, const float* WqT,
                                                           float* G, unsigned b_h, int R, unsigned has, DhArgs a) {
  // the leading scalars (operand pointers, G, sizes) are preloaded into SGPRs; the struct comes by s_load and the saved gates are
  // prefetched by `mid`, after the operand loads have been issued (stages.h head_stage_kernel)
  const int B = b_h & 0xffff, H = b_h >> 16;
  const int has_gemm = has & 1, has_gates = (has >> 1) & 1;
  __shared__ float red[2 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6;
  const int ldd = H + R;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;  // clamped: unconditional prefetch
  const size_t o = (size_t)rowc * R + col, o3 = (size_t)rowc * 3 * R + col;
  const float g0 = G[o];
  float r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f, dd = 0.f;
  auto prefetch = [&]() {
    if (has_gates) {  // wave-uniform
      r = a.rg[o]; u = a.ug[o]; n = a.ng[o]; hn = a.gh[o3 + 2 * R];
      hp = a.decin_s[(size_t)rowc * ldd + H + col];
      dd = a.ddecin_s[(size_t)rowc * ldd + H + col];
    }
  };
  float v[2] = {0.f, 0.f};
  if (has_gemm) {
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const float* const As[2] = {DP0, DQ0};
      const float* const Ws[2] = {WpT, WqT};
      const int ld[2] = {H, H}, cs[2] = {c0, c0};
      wave_gemm16_multi<NW, 2, false>(As, ld, r0, B, Ws, ld, cs, H, wave, acc, prefetch);
    }
    reduce_tiles<2, NW>(acc, red, v);
  } else {
    prefetch();
  }
  if (!own) return;
  const float g = g0 + v[0] + v[1];
  if (!has_gates) { G[o] = g; return; }
  const float dn_pre = g * (1.f - u) * (1.f - n * n);
  const float du_pre = g * (hp - n) * u * (1.f - u);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  a.dgi[o3] = dr_pre; a.dgi[o3 + R] = du_pre; a.dgi[o3 + 2 * R] = dn_pre;
  a.dgh[o3] = dr_pre; a.dgh[o3 + R] = du_pre; a.dgh[o3 + 2 * R] = dn_pre * r;
  G[o] = g * u + dd;
}

// Batches of 65 .. 256 utterances run the VRNN programs on ROW GROUPS of two row tiles (pchain_rt.h: 32-row tiles, the weight
// fragments fetched once per group).  Measured per train step on one box, [B,16000] fp32: B = 128 22.6 ms (24.7 on 16-row tiles,
// two per workgroup and link), B = 192 28.7 (36.9 as a launch per link), B = 256 38.6 (40.1).  Groups of FOUR row tiles are slower
// everywhere (B = 128: 34.8, 256: 40.7): a tile costs ~1.3 us + ~1.1 us per row tile it carries (polled fragments, MFMAs, reduction,
// epilogue stores), so fatter tiles only trade workgroups for latency; beyond 256 utterances a link has more 32-row tiles than the
// chip has workgroups for it and the launch-per-link path on 32 x 32 tiles takes over.  fp32 operands only.
// env BLVM_PCHAIN_RT_MIN_B / BLVM_PCHAIN_RT_MAX_B: the batch range (default 65 .. 256; MAX_B = 0: never); BLVM_PCHAIN_RT = 2 | 4.
inline int vrnn_rt_max_b() {
  static int v = [] {
    const char* e = getenv("BLVM_PCHAIN_RT_MAX_B");
    return e ? atoi(e) : 256;
  }();
  return v;
}
inline int vrnn_rt(int B) {
  static const int min_b = [] { const char* e = getenv("BLVM_PCHAIN_RT_MIN_B"); return e ? atoi(e) : 65; }();
  static const int forced = [] { const char* e = getenv("BLVM_PCHAIN_RT"); return e ? atoi(e) : 0; }();
  if (B < min_b || B > vrnn_rt_max_b() || pchain_max_batch() <= 0 || operand_bf16()) return 0;
  if (B <= kPchainCarveMaxB && B > pchain_max_batch()) return 0;
  return forced == 4 ? 4 : 2;
}
inline bool vrnn_row_groups(int B) { return vrnn_rt(B) > 0; }
// SHARED deal of the row-group programs: the gentle link (hidden projection; backward: the hidden-gradient product) owns no range of
// workgroups -- the chip is two halves (prior | posterior) and the gentle link runs on the posterior half right after that half's
// run, in the window in which only the prior half works (heads + phi_z run; backward: dphi + phi_z run + dz).  Pays from 7 row groups
// per link (B > 208), where a link is throughput-bound and an own range for the gentle link starves the halves; at 4..6 groups the
// own range is as good or better, and on 16-row tiles (B <= 64: a link is one hand-off latency, not tile throughput) it LOSES --
// measured r03, [64,16000]: forward 5.68 -> 7.30 ms, the three serial gentle tiles end after the prior half's run and the GRU waits.
// env BLVM_PCHAIN_SHARED = 0 | 1 overrides (any batch); BLVM_PCHAIN_RT_SHARED_TL: the row-group threshold.
inline bool vrnn_shared_deal(bool groups, int tl) {
  static const int forced = [] { const char* e = getenv("BLVM_PCHAIN_SHARED"); return e ? atoi(e) : -1; }();
  static const int rt_tl = [] { const char* e = getenv("BLVM_PCHAIN_RT_SHARED_TL"); return e ? atoi(e) : 7; }();
  if (forced >= 0) return forced != 0;
  return groups && rt_tl > 0 && tl >= rt_tl;  // (threshold 0: never)
}
inline bool vrnn_persistent(int B) { return pchain_applies(B) || vrnn_row_groups(B); }

// ---------------------------------------------------------------------------------------------------------------
// reserve / workspace carving
// ---------------------------------------------------------------------------------------------------------------
struct Reserve {
  float *P[3], *Q[3], *FZ[3], *GHb, *RG, *UG, *NG, *XQ, *XG, *RAWQ, *RAWP;
  float *Wp[3], *Wq[3], *Wph, *Wqh, *Wf[4], *Wih, *Whh;  // T16 copies of the weights the forward chain multiplies by
  // persistent forward (B <= kPchainCarveMaxB): T16 copies of every activation a link multiplies, per step [rt*16, width]
  float *H16, *P16[3], *Q16[3], *Z16, *FZ16[3], *PHI16, *x16_end;
};

size_t carve_reserve(float* base, int Tp, int B, int H, int Z, int R, Reserve* r) {
  const size_t n = (size_t)Tp * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  Reserve tmp;
  for (int i = 0; i < 3; ++i) tmp.P[i] = take(n * H);
  for (int i = 0; i < 3; ++i) tmp.Q[i] = take(n * H);
  for (int i = 0; i < 3; ++i) tmp.FZ[i] = take(n * H);
  tmp.GHb = take(n * 3 * R);
  tmp.RG = take(n * R); tmp.UG = take(n * R); tmp.NG = take(n * R);
  tmp.XQ = take(n * H);
  tmp.XG = take(n * 3 * R);
  tmp.RAWQ = take(n * Z); tmp.RAWP = take(n * Z);
  tmp.Wp[0] = take((size_t)H * R); tmp.Wq[0] = take((size_t)H * R);
  for (int i = 1; i < 3; ++i) { tmp.Wp[i] = take((size_t)H * H); tmp.Wq[i] = take((size_t)H * H); }
  tmp.Wph = take((size_t)2 * Z * H); tmp.Wqh = take((size_t)2 * Z * H);
  tmp.Wf[0] = take((size_t)H * Z);
  for (int i = 1; i < 4; ++i) tmp.Wf[i] = take((size_t)H * H);
  tmp.Wih = take((size_t)3 * R * H); tmp.Whh = take((size_t)3 * R * R);
  tmp.H16 = nullptr;
  if (B <= kPchainCarveMaxB || B <= vrnn_rt_max_b()) {
    const size_t rows = (size_t)((B + 15) / 16) * 16, m = (size_t)Tp * rows;
    tmp.H16 = take((m + rows) * R);
    for (int i = 0; i < 3; ++i) tmp.P16[i] = take(m * H);
    for (int i = 0; i < 3; ++i) tmp.Q16[i] = take(m * H);
    tmp.Z16 = take(m * Z);
    for (int i = 0; i < 3; ++i) tmp.FZ16[i] = take(m * H);
    tmp.PHI16 = take(m * H);
    tmp.x16_end = take(0);
  }
  if (r) *r = tmp;
  return off;
}

struct BwdWs {
  float *pT[3], *phT, *qT[3], *qhT, *fT[4], *wihT, *whhT;   // transposed weights, T16
  float *DGI, *DGH, *DPHI[4], *DQH, *DPH, *DP[3], *DQ[3], *G;
  // persistent backward (B <= kPchainCarveMaxB): the running state gradient as per-step slabs (written once each), [T',B,R], and
  // T16 copies of every gradient a link multiplies, per step [rt*16, width]
  float *GA, *GB, *DP16[3], *DQ16[3], *DGI16, *DGH16, *DPHI16[4], *DPH16, *DQH16, *DPHI16b, *DPHI16c, *x16_end;
  float *DPHI3b, *DPHI3c;  // row-major partial sums of DPHI[3] when its K = 3R product is split over three links (added up after the launch)
};

size_t carve_ws(float* base, int Tp, int B, int X, int H, int Z, int R, BwdWs* w) {
  (void)X;
  const size_t n = (size_t)Tp * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  BwdWs t;
  t.pT[0] = take((size_t)R * H); t.pT[1] = take((size_t)H * H); t.pT[2] = take((size_t)H * H);
  t.phT = take((size_t)H * 2 * Z);
  t.qT[0] = take((size_t)R * H); t.qT[1] = take((size_t)H * H); t.qT[2] = take((size_t)H * H);
  t.qhT = take((size_t)H * 2 * Z);
  t.fT[0] = take((size_t)Z * H);
  for (int i = 1; i < 4; ++i) t.fT[i] = take((size_t)H * H);
  t.wihT = take((size_t)H * 3 * R);
  t.whhT = take((size_t)R * 3 * R);
  t.DGI = take(n * 3 * R); t.DGH = take(n * 3 * R);
  for (int i = 0; i < 4; ++i) t.DPHI[i] = take(n * H);
  t.DQH = take(n * 2 * Z); t.DPH = take(n * 2 * Z);
  for (int i = 0; i < 3; ++i) t.DP[i] = take(n * H);
  for (int i = 0; i < 3; ++i) t.DQ[i] = take(n * H);
  t.G = take((size_t)B * R);
  t.GA = nullptr;
  if (B <= kPchainCarveMaxB || B <= vrnn_rt_max_b()) {
    const size_t m = (size_t)Tp * ((B + 15) / 16) * 16;
    t.GA = take(n * R); t.GB = take(n * R);
    for (int i = 0; i < 3; ++i) { t.DP16[i] = take(m * H); t.DQ16[i] = take(m * H); }
    t.DGI16 = take(m * 3 * R); t.DGH16 = take(m * 3 * R);
    for (int i = 0; i < 4; ++i) t.DPHI16[i] = take(m * H);
    t.DPH16 = take(m * 2 * Z); t.DQH16 = take(m * 2 * Z);
    t.DPHI16b = take(m * H); t.DPHI16c = take(m * H);
    t.x16_end = take(0);
    t.DPHI3b = take(n * H); t.DPHI3c = take(n * H);
  }
  if (w) *w = t;
  return off;
}

// A second stream for work that is independent of the recurrent chain (lazily created, one per process).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  int ensure() {
    if (stream) return BLVM_OK;
    int least = 0, greatest = 0;  // lowest priority: the batched GEMMs must not take dispatch slots from the chain's links
    BLVM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    BLVM_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, least));
    BLVM_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    BLVM_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    return BLVM_OK;
  }
};
inline SideStream& side_stream() {
  static thread_local SideStream s;
  return s;
}
// Steps per side-stream range; 0 (default) keeps the batched GEMMs behind the chain on the caller's stream.
// Measured on MI355X at [64,16000]: 24.3 ms/step without overlap, 26.1 / 25.3 / 27.2 ms with ranges of 50 / 25 / 125
// steps — the big GEMM workgroups take CU slots and memory-pipeline share from the latency-bound chain links, which
// costs more than the 2.5 ms of GEMMs it hides.  Kept as an experiment switch (env BLVM_WGRAD_OVERLAP_STEPS).
inline int overlap_chunk_steps() {
  static int v = [] {
    const char* e = getenv("BLVM_WGRAD_OVERLAP_STEPS");
    return e ? atoi(e) : 0;
  }();
  return v;
}


// a += b + c over n4 float4 (the three partial sums of DPHI[3])
__global__ __launch_bounds__(256) void add3_kernel(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 x = a[i];
    const float4 y = b[i], z = c[i];
    x.x += y.x + z.x; x.y += y.y + z.y; x.z += y.z + z.z; x.w += y.w + z.w;
    a[i] = x;
  }
}

// parts the persistent backward sequence is cut into so that finished rows' batched GEMMs overlap the rest (1 = no overlap)
inline int pchain_wgrad_parts() {
  static int v = [] {
    const char* e = getenv("BLVM_PCHAIN_WGRAD_PARTS");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// consecutive links of one shape as ONE descriptor walked inside a visit (K_LINSEQ; env BLVM_PCHAIN_LINSEQ=0: one descriptor per link)
inline bool pchain_linseq() {
  static int v = [] {
    const char* e = getenv("BLVM_PCHAIN_LINSEQ");
    return e ? atoi(e) : 1;
  }();
  return v != 0;
}

// the link in front of a run of same-shape links joins the run's descriptor visit (K_LINSEQ with its own K for the first link): the
// first prior / posterior layer (K = R, the posterior's with its x-part addend) in the forward program, the heads' gradient link
// (K = 2Z) in the backward program -- one visit less per chain and step (~1 us each, tools/probe_engine_chain.py).
// env BLVM_PCHAIN_MERGE = 0 | 1.
inline bool vrnn_merge_first(bool groups) {
  static const int v = [] { const char* e = getenv("BLVM_PCHAIN_MERGE"); return e ? atoi(e) : 1; }();
  (void)groups;  // (the row-group kernel takes the same descriptors)
  return v != 0 && pchain_linseq();
}

// the K = 3R link of the persistent backward as three K = R links (env BLVM_PCHAIN_SPLIT3=0: one link)
inline bool pchain_split3() {
  static int v = [] {
    const char* e = getenv("BLVM_PCHAIN_SPLIT3");
    return e ? atoi(e) : 1;
  }();
  return v != 0;
}

int check_dims(int Tp, int B, int X, int H, int Z, int R) {
  BLVM_REQUIRE(Tp > 0 && B > 0, "vrnn: bad Tp=%d B=%d", Tp, B);
  BLVM_REQUIRE(X > 0 && H > 0 && Z > 0 && R > 0 && X % 16 == 0 && H % 16 == 0 && Z % 16 == 0 && R % 16 == 0,
               "vrnn: X,H,Z,R must be positive multiples of 16 (got %d,%d,%d,%d)", X, H, Z, R);
  BLVM_REQUIRE(B < 65536 && H < 65536 && R < 65536, "vrnn: B, H and R must be below 65536 (packed kernel arguments)");
  return BLVM_OK;
}

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_vrnn_reserve_floats(int Tp, int B, int X, int H, int Z, int R) {
  (void)X;
  return carve_reserve(nullptr, Tp, B, H, Z, R, nullptr);
}

extern "C" size_t blvm_vrnn_bwd_workspace_floats(int Tp, int B, int X, int H, int Z, int R) {
  return carve_ws(nullptr, Tp, B, X, H, Z, R, nullptr);
}

static int vrnn_seq_fwd_impl(const BlvmVrnnWeights* w, const float* enc, const float* h0, const float* eps,
                             int Tp, int B, int X, int H, int Z, int R, int residual_posterior, float sd_eps,
                             float* decin, float* mu_q, float* sd_q, float* mu_p, float* sd_p, float* z,
                             float* reserve, hipStream_t s) {
  int rc = check_dims(Tp, B, X, H, Z, R);
  if (rc) return rc;
  BLVM_REQUIRE(w && enc && eps && decin && mu_q && sd_q && mu_p && sd_p && z && reserve, "vrnn_fwd: null pointer");
  BLVM_REQUIRE(aligned16(enc) && aligned16(decin) && aligned16(reserve) && aligned16(z),
               "vrnn_fwd: buffers must be 16-byte aligned");
  Reserve rs;
  carve_reserve(reserve, Tp, B, H, Z, R, &rs);
  const size_t n = (size_t)Tp * B;
  const int ldd = H + R;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));  // ln2 / (initial_sd - eps), initial_sd = 1

  // hoisted, state-independent halves of the two concatenated-input layers
  rc = gemm_f32(0, 0, (int)n, H, X, enc, X, w->post_w[0] + R, R + X, rs.XQ, H, w->post_b[0], 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  rc = gemm_f32(0, 0, (int)n, 3 * R, X, enc, X, w->gru_wih, X + H, rs.XG, 3 * R, w->gru_bih, 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;

  // T16 operand copies of the chain's weights (once per sequence)
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  rc = t16_pack_rows(w->prior_w[0], R, H, R, rs.Wp[0], s); if (rc) return rc;
  rc = t16_pack_rows(w->post_w[0], R + X, H, R, rs.Wq[0], s); if (rc) return rc;  // the h columns
  for (int l = 1; l < 3; ++l) {
    rc = t16_pack_rows(w->prior_w[l], H, H, H, rs.Wp[l], s); if (rc) return rc;
    rc = t16_pack_rows(w->post_w[l], H, H, H, rs.Wq[l], s); if (rc) return rc;
  }
  rc = t16_pack_rows(w->prior_hw, H, 2 * Z, H, rs.Wph, s); if (rc) return rc;
  rc = t16_pack_rows(w->post_hw, H, 2 * Z, H, rs.Wqh, s); if (rc) return rc;
  rc = t16_pack_rows(w->phi_w[0], Z, H, Z, rs.Wf[0], s); if (rc) return rc;
  for (int l = 1; l < 4; ++l) { rc = t16_pack_rows(w->phi_w[l], H, H, H, rs.Wf[l], s); if (rc) return rc; }
  rc = t16_pack_rows(w->gru_wih + X, X + H, 3 * R, H, rs.Wih, s); if (rc) return rc;  // the phi columns
  rc = t16_pack_rows(w->gru_whh, R, 3 * R, R, rs.Whh, s); if (rc) return rc;
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;

  // initial state -> h-part of decin row 0
  if (h0) BLVM_HIP(hipMemcpy2DAsync(decin + H, sizeof(float) * ldd, h0, sizeof(float) * R, sizeof(float) * R, B, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemset2DAsync(decin + H, sizeof(float) * ldd, 0, sizeof(float) * R, B, s));

  const int rt = (B + 15) / 16;
  if (vrnn_persistent(B) && device_cus() >= 32) {
    // Persistent path (pchain.h / pchain.hip): the nine links of a step as a program of 13 descriptors, one launch for the whole
    // sequence.  Links on the critical path share the workgroups [0, g); the GRU's hidden projection gh_t = h_{t-1} W_hh^T + b_hh
    // (3R columns, first needed by the GRU link's epilogue eight links later) has its own range behind them and polls gently.
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, ctR = R / 16, cus = device_cus() & ~7;
    const long sH = (long)B * H, sZ = (long)B * Z, sR = (long)B * R, s3R = 3 * sR, sD = (long)B * ldd;
    const long xH = (long)rt * 16 * H, xZ = (long)rt * 16 * Z, xR = (long)rt * 16 * R;
    // tiles of a link per column tile: row tiles, or (65 <= B <= 256) groups of row tiles (pchain_rt.h)
    const bool groups = vrnn_row_groups(B);
    const int RTG = groups ? vrnn_rt(B) : 1;
    const int tl = (rt + RTG - 1) / RTG;
    // hidden projection: up to a quarter of the chip (groups of four row tiles: up to half)
    // shared deal (row groups, many row tiles): no range of its own for the hidden projection -- it runs on the posterior half after that
    // half's run, in the window where only the prior half works on the heads and the phi_z run
    const bool shared = vrnn_shared_deal(groups, tl);
    const int def_n = shared ? 0 : range_for(3 * ctR * tl, std::min(RTG >= 4 ? cus / 2 : cus / 4, RTG >= 4 ? 128 : 64));
    const int half = range_for(ctH * tl, (cus - def_n) / 2);            // prior | posterior halves of a link
    const int g = 2 * half;
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.rt_group = RTG;
    bld.p.S = Tp; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = groups ? 8 : 4;
    bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = g;
    auto lin = [&](const float* A16, long a_step, const float* W, int K, const float* bias, const float* add, long add_step, int ldadd, float* orm,
                   long rm_step, int ldo, float* o16, long o16_step, int n16, int ct, int wg0, int nwg, int flags) {
      Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags, 0, Tp);
      bld.ptr(d, 0, A16, a_step); bld.ptr(d, 1, W); bld.ptr(d, 2, bias); bld.ptr(d, 3, add, add_step);
      bld.ptr(d, 5, orm, rm_step); bld.ptr(d, 6, o16, o16_step);
      d.ld[1] = ldadd; d.ld[3] = ldo; d.n16[0] = n16; d.f[0] = 0.f;
    };
    // F1: first prior layer | h-half of the first posterior layer | hidden projection
    const bool merge1 = vrnn_merge_first(groups);  // (then the first prior / posterior layer opens the run below)
    if (!merge1) {
      lin(rs.H16, xR, rs.Wp[0], R, w->prior_b[0], nullptr, 0, 0, rs.P[0], sH, H, rs.P16[0], xH, ctH, ctH, 0, half, DF_RELU);
      lin(rs.H16, xR, rs.Wq[0], R, nullptr, rs.XQ, sH, H, rs.Q[0], sH, H, rs.Q16[0], xH, ctH, ctH, half, half, DF_RELU);
    }
    auto hproj = [&](int wg0, int nwg) {
      lin(rs.H16, xR, rs.Whh, R, w->gru_bhh, nullptr, 0, 0, rs.GHb, s3R, 3 * R, nullptr, 0, 0, 3 * ctR, wg0, nwg,
          DF_RM_SC1 | DF_GENTLE | ((pchain_tune() & 16) ? DF_CANARY : 0));
    };
    if (!shared) hproj(g, def_n);
    // a run of consecutive links of one shape as one descriptor: out_i = relu(A_i W_i^T + b_i), A_{i+1} = out_i
    struct SeqLink { const float* W; const float* bias; float* orm; long rm_step; int ldo; float* o16; };
    auto linseq = [&](const float* A16, long a_step, int K, int n, const SeqLink* L, int wg0, int nwg, int K0 = 0, const float* add0 = nullptr,
                      long add0_step = 0, int ldadd0 = 0) {
      Desc& d = bld.add(K_LINSEQ, ctH, wg0, nwg, K, DF_RELU, 0, Tp);
      bld.ptr(d, 0, A16, a_step); bld.ptr(d, 17, add0, add0_step);
      d.i[3] = K0 != K ? K0 : 0; d.i[0] = ldadd0;
      for (int i = 0; i < n; ++i) {
        bld.ptr(d, 1 + i, L[i].W); bld.ptr(d, 5 + i, L[i].bias); bld.ptr(d, 9 + i, L[i].orm, L[i].rm_step); bld.ptr(d, 13 + i, L[i].o16, xH);
        d.ld[i] = L[i].ldo;
      }
      d.n16[0] = ctH; d.i[1] = n; d.f[0] = 0.f;
    };
    const bool seq = pchain_linseq();
    // F2, F3
    if (seq && merge1) {  // F1 .. F3 of a chain: one visit
      const SeqLink lp[3] = {{rs.Wp[0], w->prior_b[0], rs.P[0], sH, H, rs.P16[0]}, {rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      const SeqLink lq[3] = {{rs.Wq[0], nullptr, rs.Q[0], sH, H, rs.Q16[0]}, {rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      linseq(rs.H16, xR, H, 3, lp, 0, half, R);
      linseq(rs.H16, xR, H, 3, lq, half, half, R, rs.XQ, sH, H);
    } else if (seq) {
      const SeqLink lp[2] = {{rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      const SeqLink lq[2] = {{rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      linseq(rs.P16[0], xH, H, 2, lp, 0, half);
      linseq(rs.Q16[0], xH, H, 2, lq, half, half);
    } else {
      for (int l = 1; l < 3; ++l) {
        lin(rs.P16[l - 1], xH, rs.Wp[l], H, w->prior_b[l], nullptr, 0, 0, rs.P[l], sH, H, rs.P16[l], xH, ctH, ctH, 0, half, DF_RELU);
        lin(rs.Q16[l - 1], xH, rs.Wq[l], H, w->post_b[l], nullptr, 0, 0, rs.Q[l], sH, H, rs.Q16[l], xH, ctH, ctH, half, half, DF_RELU);
      }
    }
    if (shared) hproj(half, half);
    {  // F4: heads + sample
      Desc& d = bld.add(K_HEAD, ctZ, 0, range_for(ctZ * tl, shared ? half : g), H, 0, 0, Tp);
      bld.ptr(d, 0, rs.P16[2], xH); bld.ptr(d, 1, rs.Q16[2], xH); bld.ptr(d, 2, rs.Wph); bld.ptr(d, 3, w->prior_hb); bld.ptr(d, 4, rs.Wqh);
      bld.ptr(d, 5, w->post_hb); bld.ptr(d, 6, eps, sZ); bld.ptr(d, 7, mu_p, sZ); bld.ptr(d, 8, sd_p, sZ); bld.ptr(d, 9, mu_q, sZ);
      bld.ptr(d, 10, sd_q, sZ); bld.ptr(d, 11, rs.RAWP, sZ); bld.ptr(d, 12, rs.RAWQ, sZ); bld.ptr(d, 13, nullptr);
      bld.ptr(d, 14, z, sZ); bld.ptr(d, 15, rs.Z16, xZ);
      d.ld[3] = Z; d.n16[0] = ctZ; d.i[0] = Z; d.i[1] = residual_posterior; d.f[0] = beta; d.f[1] = 1.f / beta; d.f[2] = sd_eps;
    }
    // F5..F8: phi_z MLP (the last layer writes phi into decin row t)
    const int first_seq = !seq ? 4 : (Z == H ? 0 : 1);  // (the first layer's K is Z: part of the run only when Z == H)
    for (int l = 0; l < first_seq; ++l) {
      const float* A = l == 0 ? rs.Z16 : rs.FZ16[l - 1];
      lin(A, l == 0 ? xZ : xH, rs.Wf[l], l == 0 ? Z : H, w->phi_b[l], nullptr, 0, 0, l == 3 ? decin : rs.FZ[l], l == 3 ? sD : sH, l == 3 ? ldd : H,
          l == 3 ? rs.PHI16 : rs.FZ16[l], xH, ctH, ctH, 0, range_for(ctH * tl, shared ? half : g), DF_RELU);
    }
    if (first_seq < 4) {
      SeqLink lf[4];
      for (int l = first_seq; l < 4; ++l)
        lf[l - first_seq] = SeqLink{rs.Wf[l], w->phi_b[l], l == 3 ? decin : rs.FZ[l], l == 3 ? sD : sH, l == 3 ? ldd : H, l == 3 ? rs.PHI16 : rs.FZ16[l]};
      linseq(first_seq == 0 ? rs.Z16 : rs.FZ16[first_seq - 1], first_seq == 0 ? xZ : xH, H, 4 - first_seq, lf, 0, range_for(ctH * tl, shared ? half : g));
    }
    {  // F9: GRU
      Desc& d = bld.add(K_GRU, ctR, 0, range_for(ctR * tl, g), H, 0, 0, Tp);
      bld.ptr(d, 0, rs.PHI16, xH); bld.ptr(d, 1, rs.Wih); bld.ptr(d, 2, rs.XG, s3R); bld.ptr(d, 3, rs.GHb, s3R); bld.ptr(d, 4, decin + H, sD);
      bld.ptr(d, 5, decin + sD + H, sD); bld.ptr(d, 6, rs.H16 + xR, xR); bld.ptr(d, 7, rs.RG, sR); bld.ptr(d, 8, rs.UG, sR); bld.ptr(d, 9, rs.NG, sR);
      d.ld[0] = ldd; d.ld[3] = ldd; d.n16[0] = ctR; d.i[0] = R;
    }
    BLVM_REQUIRE(!bld.overflow, "vrnn_fwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    // sentinel-fill what the launch polls: the T16 copies, the hidden projection, and decin (the GRU link polls words of h)
    BLVM_HIP(pchain_fill_sentinel(rs.H16, (size_t)(reinterpret_cast<char*>(rs.x16_end) - reinterpret_cast<char*>(rs.H16)), s));
    BLVM_HIP(pchain_fill_sentinel(rs.GHb, sizeof(float) * n * 3 * R, s));
    BLVM_HIP(pchain_fill_sentinel(decin, sizeof(float) * n * ldd, s));  // rows 0..T'-1; row T' only receives h_n
    if (h0) BLVM_HIP(hipMemcpy2DAsync(decin + H, sizeof(float) * ldd, h0, sizeof(float) * R, sizeof(float) * R, B, hipMemcpyDeviceToDevice, s));
    else BLVM_HIP(hipMemset2DAsync(decin + H, sizeof(float) * ldd, 0, sizeof(float) * R, B, s));
    rc = pchain_rows_to_t16(h0, R, B, R, rs.H16, s);
    if (rc) return rc;
    return pchain_launch(bld.p, s);
  }
  for (int t = 0; t < Tp; ++t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z, oR = (size_t)t * B * R, o3R = (size_t)t * B * 3 * R;
    const float* dec_t = decin + (size_t)t * B * ldd;
    float* dec_n = decin + (size_t)(t + 1) * B * ldd;
    const float* hprev = dec_t + H;
    LinLaunch a;
    a.B = B;
    // F1: first prior layer | h-half of first posterior layer | GRU hidden projection
    a.nseg = 3;
    a.seg[0] = seg(hprev, ldd, rs.Wp[0], R, w->prior_b[0], nullptr, 0, nullptr, 0, rs.P[0] + oH, H, H, R, 1);
    a.seg[1] = seg(hprev, ldd, rs.Wq[0], R, nullptr, rs.XQ + oH, H, nullptr, 0, rs.Q[0] + oH, H, H, R, 1);
    a.seg[2] = seg(hprev, ldd, rs.Whh, R, w->gru_bhh, nullptr, 0, nullptr, 0, rs.GHb + o3R, 3 * R, 3 * R, R, 0);
    launch_lin(a, s);
    // F2, F3
    a.nseg = 2;
    for (int l = 1; l < 3; ++l) {
      a.seg[0] = seg(rs.P[l - 1] + oH, H, rs.Wp[l], H, w->prior_b[l], nullptr, 0, nullptr, 0, rs.P[l] + oH, H, H, H, 1);
      a.seg[1] = seg(rs.Q[l - 1] + oH, H, rs.Wq[l], H, w->post_b[l], nullptr, 0, nullptr, 0, rs.Q[l] + oH, H, H, H, 1);
      launch_lin(a, s);
    }
    // F4: heads + sample
    HeadArgs h;
    h.P = rs.P[2] + oH; h.Q = rs.Q[2] + oH;
    h.Wp = rs.Wph; h.bp = w->prior_hb; h.Wq = rs.Wqh; h.bq = w->post_hb;
    h.eps = eps + oZ;
    h.mu_p = mu_p + oZ; h.sd_p = sd_p + oZ; h.mu_q = mu_q + oZ; h.sd_q = sd_q + oZ; h.z = z + oZ;
    h.raw_p = rs.RAWP + oZ; h.raw_q = rs.RAWQ + oZ;
    h.B = B; h.H = H; h.Z = Z; h.residual = residual_posterior;
    h.beta = beta; h.inv_beta = 1.f / beta; h.sd_eps = sd_eps; h.muq_raw = nullptr;
    launch_head(h, pick_nw(H, 4), dim3(Z / 16, rt), s);
    // F5..F8: phi_z MLP (last layer writes phi into decin row t)
    a.nseg = 1;
    a.seg[0] = seg(z + oZ, Z, rs.Wf[0], Z, w->phi_b[0], nullptr, 0, nullptr, 0, rs.FZ[0] + oH, H, H, Z, 1);
    launch_lin(a, s);
    a.seg[0] = seg(rs.FZ[0] + oH, H, rs.Wf[1], H, w->phi_b[1], nullptr, 0, nullptr, 0, rs.FZ[1] + oH, H, H, H, 1);
    launch_lin(a, s);
    a.seg[0] = seg(rs.FZ[1] + oH, H, rs.Wf[2], H, w->phi_b[2], nullptr, 0, nullptr, 0, rs.FZ[2] + oH, H, H, H, 1);
    launch_lin(a, s);
    a.seg[0] = seg(rs.FZ[2] + oH, H, rs.Wf[3], H, w->phi_b[3], nullptr, 0, nullptr, 0, decin + (size_t)t * B * ldd, ldd, H, H, 1);
    launch_lin(a, s);
    // F9: GRU
    {
      const int nw = pick_nw(H, 3);
      const dim3 grid(R / 16, rt);
      const float *xg_t = rs.XG + o3R, *gh_t = rs.GHb + o3R;
      float *rg_t = rs.RG + oR, *ug_t = rs.UG + oR, *ng_t = rs.NG + oR;
      if (nw == 16) hipLaunchKernelGGL((gru_stage_kernel<16>), grid, dim3(1024), 0, s, dec_t, (const float*)rs.Wih, xg_t, gh_t, B, H, R, dec_n, rg_t, ug_t, ng_t);
      else if (nw == 8) hipLaunchKernelGGL((gru_stage_kernel<8>), grid, dim3(512), 0, s, dec_t, (const float*)rs.Wih, xg_t, gh_t, B, H, R, dec_n, rg_t, ug_t, ng_t);
      else hipLaunchKernelGGL((gru_stage_kernel<4>), grid, dim3(256), 0, s, dec_t, (const float*)rs.Wih, xg_t, gh_t, B, H, R, dec_n, rg_t, ug_t, ng_t);
    }
  }
  BLVM_CHECK_LAUNCH("vrnn_seq_fwd");
  return BLVM_OK;
}

extern "C" int blvm_vrnn_seq_fwd(const BlvmVrnnWeights* w, const float* enc, const float* h0, const float* eps,
                                 int Tp, int B, int X, int H, int Z, int R, int residual_posterior, float sd_eps,
                                 float* decin, float* mu_q, float* sd_q, float* mu_p, float* sd_p, float* z,
                                 float* reserve, void* stream_) {
  BLVM_REQUIRE(w != nullptr, "vrnn_fwd: null pointer");
  ChainKey key("vrnn_fwd");
  key.add(*w).add(enc).add(h0).add(eps).add(Tp).add(B).add(X).add(H).add(Z).add(R).add(residual_posterior).add(sd_eps);
  key.add(decin).add(mu_q).add(sd_q).add(mu_p).add(sd_p).add(z).add(reserve);
  return run_chain(key, static_cast<hipStream_t>(stream_), [&](hipStream_t s) {
    return vrnn_seq_fwd_impl(w, enc, h0, eps, Tp, B, X, H, Z, R, residual_posterior, sd_eps, decin, mu_q, sd_q, mu_p, sd_p, z, reserve, s);
  });
}

static int vrnn_seq_bwd_impl(const BlvmVrnnWeights* w, const float* enc, const float* eps, const float* decin,
                             const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p,
                             const float* z, const float* reserve, const float* d_decin, const int32_t* x_sl,
                             const float* c_raw, const float* c_fn, int stride, float fn_floor, int Tp, int B, int X, int H, int Z,
                             int R, int residual_posterior, float sd_eps, float* d_enc, float* d_h0,
                             const BlvmVrnnGrads* gr, float* workspace, hipStream_t s) {
  int rc = check_dims(Tp, B, X, H, Z, R);
  if (rc) return rc;
  BLVM_REQUIRE(w && enc && eps && decin && mu_q && sd_q && mu_p && sd_p && z && reserve && d_decin && workspace && gr,
               "vrnn_bwd: null pointer");
  BLVM_REQUIRE((c_fn == nullptr && c_raw == nullptr) || x_sl != nullptr, "vrnn_bwd: KL coefficients need x_sl");
  BLVM_REQUIRE(aligned16(workspace) && aligned16(reserve) && aligned16(d_decin), "vrnn_bwd: buffers must be 16-byte aligned");
  Reserve rs;
  carve_reserve(const_cast<float*>(reserve), Tp, B, H, Z, R, &rs);
  BwdWs ws;
  carve_ws(workspace, Tp, B, X, H, Z, R, &ws);
  const size_t n = (size_t)Tp * B;
  const int ldd = H + R;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));

  // transposed T16 operand copies of every weight the chain multiplies from the right
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  rc = t16_pack_transposed(w->prior_w[0], R, H, R, ws.pT[0], s); if (rc) return rc;
  rc = t16_pack_transposed(w->prior_w[1], H, H, H, ws.pT[1], s); if (rc) return rc;
  rc = t16_pack_transposed(w->prior_w[2], H, H, H, ws.pT[2], s); if (rc) return rc;
  rc = t16_pack_transposed(w->prior_hw, H, 2 * Z, H, ws.phT, s); if (rc) return rc;
  rc = t16_pack_transposed(w->post_w[0], R + X, H, R, ws.qT[0], s); if (rc) return rc;
  rc = t16_pack_transposed(w->post_w[1], H, H, H, ws.qT[1], s); if (rc) return rc;
  rc = t16_pack_transposed(w->post_w[2], H, H, H, ws.qT[2], s); if (rc) return rc;
  rc = t16_pack_transposed(w->post_hw, H, 2 * Z, H, ws.qhT, s); if (rc) return rc;
  rc = t16_pack_transposed(w->phi_w[0], Z, H, Z, ws.fT[0], s); if (rc) return rc;
  for (int i = 1; i < 4; ++i) { rc = t16_pack_transposed(w->phi_w[i], H, H, H, ws.fT[i], s); if (rc) return rc; }
  rc = t16_pack_transposed(w->gru_wih + X, X + H, 3 * R, H, ws.wihT, s); if (rc) return rc;
  rc = t16_pack_transposed(w->gru_whh, R, 3 * R, R, ws.whhT, s); if (rc) return rc;
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;

  BLVM_HIP(hipMemsetAsync(ws.G, 0, sizeof(float) * (size_t)B * R, s));
  const int rt = (B + 15) / 16;

  auto launch_dh = [&](int t_gemm, int s_gates) {
    DhArgs d;
    d.has_gemm = t_gemm >= 0; d.has_gates = s_gates >= 0;
    d.DP0 = d.has_gemm ? ws.DP[0] + (size_t)t_gemm * B * H : nullptr;
    d.DQ0 = d.has_gemm ? ws.DQ[0] + (size_t)t_gemm * B * H : nullptr;
    d.WpT = ws.pT[0]; d.WqT = ws.qT[0]; d.G = ws.G;
    const size_t sg = d.has_gates ? (size_t)s_gates : 0;
    d.rg = rs.RG + sg * B * R; d.ug = rs.UG + sg * B * R; d.ng = rs.NG + sg * B * R;
    d.gh = rs.GHb + sg * B * 3 * R;
    d.decin_s = decin + sg * B * ldd; d.ddecin_s = d_decin + sg * B * ldd;
    d.dgi = ws.DGI + sg * B * 3 * R; d.dgh = ws.DGH + sg * B * 3 * R;
    d.B = B; d.H = H; d.R = R;
    {
      const int nw = pick_nw(H, 2);
      const dim3 grid(R / 16, rt);
      const unsigned b_h = (unsigned)B | ((unsigned)H << 16), has = (d.has_gemm ? 1u : 0u) | (d.has_gates ? 2u : 0u);
      if (nw == 16) hipLaunchKernelGGL((dh_stage_kernel<16>), grid, dim3(1024), 0, s, d.DP0, d.DQ0, d.WpT, d.WqT, d.G, b_h, R, has, d);
      else if (nw == 8) hipLaunchKernelGGL((dh_stage_kernel<8>), grid, dim3(512), 0, s, d.DP0, d.DQ0, d.WpT, d.WqT, d.G, b_h, R, has, d);
      else hipLaunchKernelGGL((dh_stage_kernel<4>), grid, dim3(256), 0, s, d.DP0, d.DQ0, d.WpT, d.WqT, d.G, b_h, R, has, d);
    }
  };

  // ---- batched, state-independent part: d(enc) and every weight gradient as large MFMA GEMMs over a row range ----
  // Rows [r0, r0+nr) of every per-step gradient buffer are final once the chain has passed step r0/B, so the range
  // can be processed on a second stream UNDER the latency-bound BPTT chain (which leaves most CUs idle).
  const float* hprev_all = decin + H;  // [n rows, ld = H+R]
  auto batched = [&](size_t r0, size_t nr, hipStream_t st) -> int {
    int rc2 = BLVM_OK;
#define TRY(x) do { rc2 = (x); if (rc2) return rc2; } while (0)
    const float* DGI = ws.DGI + r0 * 3 * R; const float* DGH = ws.DGH + r0 * 3 * R;
    const float* encr = enc + r0 * X; const float* decr = decin + r0 * ldd; const float* hpr = hprev_all + r0 * ldd;
    if (d_enc) {
      TRY(gemm_f32(0, 1, (int)nr, X, H, ws.DQ[0] + r0 * H, H, w->post_w[0] + R, R + X, d_enc + r0 * X, X, nullptr, 0, 0.f, nullptr, 0, 0, 1, st));
      TRY(gemm_f32(0, 1, (int)nr, X, 3 * R, DGI, 3 * R, w->gru_wih, X + H, d_enc + r0 * X, X, nullptr, 0, 0.f, nullptr, 0, 1, 1, st));
    }
    // every weight gradient of the chain as ONE grouped launch (gemm.hip gemm_wgrad_group; D [rows, M] x Act [rows, N] -> dW [M, N], db [M])
    {
      WgradGroup grp;
      auto job = [&](const float* D, int ldd_, int n_out, const float* Act, int lda, int k_in, float* dW, int ldw, float* db = nullptr) {
        grp.add(D, ldd_, n_out, Act, lda, k_in, dW, ldw, db);
      };
      job(DGI, 3 * R, 3 * R, encr, X, X, gr->gru_wih, X + H, gr->gru_bih);
      job(DGI, 3 * R, 3 * R, decr, ldd, H, gr->gru_wih ? gr->gru_wih + X : nullptr, X + H);
      job(DGH, 3 * R, 3 * R, hpr, ldd, R, gr->gru_whh, R, gr->gru_bhh);
      job(ws.DPHI[0] + r0 * H, H, H, z + r0 * Z, Z, Z, gr->phi_w[0], Z, gr->phi_b[0]);
      for (int l = 1; l < 4; ++l) job(ws.DPHI[l] + r0 * H, H, H, rs.FZ[l - 1] + r0 * H, H, H, gr->phi_w[l], H, gr->phi_b[l]);
      job(ws.DPH + r0 * 2 * Z, 2 * Z, 2 * Z, rs.P[2] + r0 * H, H, H, gr->prior_hw, H, gr->prior_hb);
      job(ws.DQH + r0 * 2 * Z, 2 * Z, 2 * Z, rs.Q[2] + r0 * H, H, H, gr->post_hw, H, gr->post_hb);
      for (int l = 2; l >= 1; --l) {
        job(ws.DP[l] + r0 * H, H, H, rs.P[l - 1] + r0 * H, H, H, gr->prior_w[l], H, gr->prior_b[l]);
        job(ws.DQ[l] + r0 * H, H, H, rs.Q[l - 1] + r0 * H, H, H, gr->post_w[l], H, gr->post_b[l]);
      }
      job(ws.DP[0] + r0 * H, H, H, hpr, ldd, R, gr->prior_w[0], R, gr->prior_b[0]);
      job(ws.DQ[0] + r0 * H, H, H, hpr, ldd, R, gr->post_w[0], R + X);
      job(ws.DQ[0] + r0 * H, H, H, encr, X, X, gr->post_w[0] ? gr->post_w[0] + R : nullptr, R + X, gr->post_b[0]);
      TRY(grp.run(nr, st));
    }
#undef TRY
    return BLVM_OK;
  };
  if (vrnn_persistent(B) && device_cus() >= 32) {
    // Persistent path: the whole BPTT chain as a program of 13 descriptors walked for s = 0 .. T' (t = T'-1-s: last-step slabs and
    // negative strides), then the batched weight-gradient GEMMs.  The running gradient wrt the recurrent state lives in per-step
    // slabs so that every location is written once: GA[t] = g_t * u_t + decoder gradient (written by the GRU-backward link),
    // GB[t] = GA[t] + DGH[t] W_hh (a K = 3R product nothing needs before the NEXT step's GRU-backward link: own range, gentle polls).
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, ctR = R / 16, cus = device_cus() & ~7, T = Tp;
    const long sH = (long)B * H, sZ = (long)B * Z, sR = (long)B * R, s3R = 3 * sR, s2Z = 2 * sZ, sD = (long)B * ldd;
    const long xH = (long)rt * 16 * H, x2Z = (long)rt * 16 * 2 * Z, x3R = (long)rt * 16 * 3 * R;
    const bool groups = vrnn_row_groups(B);
    const int RTG = groups ? vrnn_rt(B) : 1;
    const int tl = (rt + RTG - 1) / RTG;  // tiles of a link per column tile: row tiles, or groups of them (pchain_rt.h)
    // (shared deal: see vrnn_fwd -- here the gentle link is GB, which the posterior half runs while the prior half
    // takes the gradient back through the phi_z run and the heads)
    const bool shared = vrnn_shared_deal(groups, tl);
    const int def_n = shared ? 0 : range_for(ctR * tl, std::min(cus / 4, 64));  // GB link
    const int half = range_for(ctH * tl, (cus - def_n) / 2), g = 2 * half;
    const int wide = shared ? half : g;  // range of the links between the GRU backward and the heads
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.rt_group = RTG;
    bld.p.S = T + 1; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = groups ? 8 : 2;
    bld.p.prof = pchain_profile_buffer() ? pchain_profile_buffer() + 64 : nullptr; bld.p.prof_wg = g;
    auto last = [&](const float* base, long step) { return base ? base + (long)(T - 1) * step : nullptr; };  // slab of t = T'-1
    {  // Ba: complete the gradient wrt h_t, GRU gate derivatives of step t (s = T': only the gradient wrt the initial state)
      Desc& d = bld.add(K_GRUB, ctR, 0, range_for(ctR * tl, g), H, 0, 0, T + 1);
      bld.ptr(d, 0, ws.DP16[0] + (long)T * xH, -xH); bld.ptr(d, 1, ws.DQ16[0] + (long)T * xH, -xH); bld.ptr(d, 2, ws.pT[0]); bld.ptr(d, 3, ws.qT[0]);
      bld.ptr(d, 4, ws.GB + (long)T * sR, -sR);
      bld.ptr(d, 5, last(rs.RG, sR), -sR); bld.ptr(d, 6, last(rs.UG, sR), -sR); bld.ptr(d, 7, last(rs.NG, sR), -sR); bld.ptr(d, 8, last(rs.GHb, s3R), -s3R);
      bld.ptr(d, 9, last(decin, sD) + H, -sD); bld.ptr(d, 10, last(d_decin, sD) + H, -sD);
      bld.ptr(d, 11, last(ws.DGI, s3R), -s3R); bld.ptr(d, 12, last(ws.DGI16, x3R), -x3R); bld.ptr(d, 13, last(ws.DGH, s3R), -s3R);
      bld.ptr(d, 14, last(ws.DGH16, x3R), -x3R); bld.ptr(d, 15, last(ws.GA, sR), -sR); bld.ptr(d, 16, d_h0 ? d_h0 : ws.G);
      d.ld[0] = ldd; d.ld[3] = 3 * R; d.n16[0] = 3 * ctR; d.i[0] = R; d.i[1] = 1; d.i[2] = T; d.i[3] = 1;
    }
    auto lin = [&](const float* A16, long a_x, const float* W, int K, const float* add, long add_step, int ldadd, const float* gate, long gate_step,
                   int ldgate, float* orm, long rm_step, int ldo, float* o16, long o16_x, int n16, int ct, int wg0, int nwg, int flags) {
      Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags, 0, T);
      bld.ptr(d, 0, last(A16, a_x), -a_x); bld.ptr(d, 1, W); bld.ptr(d, 3, last(add, add_step), -add_step); bld.ptr(d, 4, last(gate, gate_step), -gate_step);
      bld.ptr(d, 5, last(orm, rm_step), -rm_step); bld.ptr(d, 6, last(o16, o16_x), -o16_x);
      d.ld[1] = ldadd; d.ld[2] = ldgate; d.ld[3] = ldo; d.n16[0] = n16; d.f[0] = 0.f;
    };
    // Bb: dphi through the GRU input projection (+ the decoder's gradient, through phi's ReLU) | GB[t] = GA[t] + DGH[t] W_hh
    // The K = 3R product is the fattest link of the step (96 KB of operands per tile): when a third range of workgroups is free it
    // runs as three K = R links side by side (the r | u | n thirds of DGI and of W_ih^T), each writing a full slab of partial sums
    // (the derivative mask distributes over the sum; the decoder's gradient joins the first), and B3 adds the three slabs up as it
    // loads them.
    const int spare = cus - g - def_n;
    const bool split3 = !shared && spare >= 8 && half >= 8 && pchain_split3();
    if (split3) {
      const size_t wthird = (size_t)ctR * 256 / (bld.p.bf16 ? 2 : 1);  // the packed weight's k-chunks [ctR * part, ...) (bf16 packs: half the floats)
      float* const orm[3] = {ws.DPHI[3], ws.DPHI3b, ws.DPHI3c};
      float* const o16[3] = {ws.DPHI16[3], ws.DPHI16b, ws.DPHI16c};
      const int wg0s[3] = {0, half, g + def_n}, nwgs[3] = {range_for(ctH * tl, half), range_for(ctH * tl, half), range_for(ctH * tl, spare)};
      for (int part = 0; part < 3; ++part) {
        lin(ws.DGI16 + (size_t)part * ctR * 256, x3R, ws.wihT + part * wthird, R, part == 0 ? d_decin : nullptr, sD, ldd, decin, sD, ldd, orm[part], sH, H, o16[part], xH,
            ctH, ctH, wg0s[part], nwgs[part], 0);
        Desc& d = bld.p.d[bld.p.ndesc - 1];
        d.ld[0] = 3 * R; d.i[0] = 3 * R;  // widths of the slab / of the packed rows the K-range is taken from
      }
    } else {
      lin(ws.DGI16, x3R, ws.wihT, 3 * R, d_decin, sD, ldd, decin, sD, ldd, ws.DPHI[3], sH, H, ws.DPHI16[3], xH, ctH, ctH, 0, range_for(ctH * tl, wide), 0);
    }
    lin(ws.DGH16, x3R, ws.whhT, 3 * R, ws.GA, sR, R, nullptr, 0, 0, ws.GB, sR, R, nullptr, 0, 0, ctR, shared ? half : g, shared ? half : def_n,
        DF_ADD_POLLED | DF_RM_SC1 | DF_GENTLE | ((pchain_tune() & 16) ? DF_CANARY : 0));
    // B3..B5: back through phi_z layers 3, 2, 1
    // a run of consecutive backward links of one shape as one descriptor (K_LINSEQ): D_{i+1} = (D_i W_i) masked by the saved activation
    struct SeqLinkB { const float* WT; const float* gate; float* orm; float* o16; };
    auto linseq_b = [&](const float* A16, int n, const SeqLinkB* L, int wg0, int nwg, int flags, int K0 = 0, long a_x = 0) -> Desc& {
      Desc& d = bld.add(K_LINSEQ, ctH, wg0, nwg, H, flags | DF_SEQ_GATE, 0, T);
      if (a_x == 0) a_x = xH;
      bld.ptr(d, 0, last(A16, a_x), -a_x);
      d.i[3] = (K0 != 0 && K0 != H) ? K0 : 0;  // (the first link's own K: the link in front of the run, in the run's visit)
      for (int i = 0; i < n; ++i) {
        bld.ptr(d, 1 + i, L[i].WT); bld.ptr(d, 5 + i, last(L[i].gate, sH), -sH); bld.ptr(d, 9 + i, last(L[i].orm, sH), -sH);
        bld.ptr(d, 13 + i, last(L[i].o16, xH), -xH);
        d.ld[i] = H;
      }
      d.n16[0] = ctH; d.i[1] = n; d.i[2] = H; d.f[0] = 0.f;
      return d;
    };
    const bool seq = pchain_linseq();
    if (seq) {
      const SeqLinkB lf[3] = {{ws.fT[3], rs.FZ[2], ws.DPHI[2], ws.DPHI16[2]}, {ws.fT[2], rs.FZ[1], ws.DPHI[1], ws.DPHI16[1]}, {ws.fT[1], rs.FZ[0], ws.DPHI[0], ws.DPHI16[0]}};
      if (split3) {  // the link that adds the three partial-sum slabs up is a K_LIN of its own (a run's links are plain), the other two a run
        lin(ws.DPHI16[3], xH, ws.fT[3], H, nullptr, 0, 0, rs.FZ[2], sH, H, ws.DPHI[2], sH, H, ws.DPHI16[2], xH, ctH, ctH, 0, range_for(ctH * tl, wide), DF_A_SUM3);
        Desc& d = bld.p.d[bld.p.ndesc - 1];
        bld.ptr(d, 8, ws.DPHI16b + (long)(T - 1) * xH, -xH); bld.ptr(d, 9, ws.DPHI16c + (long)(T - 1) * xH, -xH);
        linseq_b(ws.DPHI16[2], 2, lf + 1, 0, range_for(ctH * tl, wide), 0);
      } else {
        linseq_b(ws.DPHI16[3], 3, lf, 0, range_for(ctH * tl, wide), 0);
      }
    } else {
      for (int l = 3; l >= 1; --l) {
        lin(ws.DPHI16[l], xH, ws.fT[l], H, nullptr, 0, 0, rs.FZ[l - 1], sH, H, ws.DPHI[l - 1], sH, H, ws.DPHI16[l - 1], xH, ctH, ctH, 0, range_for(ctH * tl, wide),
            l == 3 && split3 ? DF_A_SUM3 : 0);
        if (l == 3 && split3) {
          Desc& d = bld.p.d[bld.p.ndesc - 1];
          bld.ptr(d, 8, ws.DPHI16b + (long)(T - 1) * xH, -xH); bld.ptr(d, 9, ws.DPHI16c + (long)(T - 1) * xH, -xH);
        }
      }
    }
    {  // B6: dz and the heads
      Desc& d = bld.add(K_DZ, ctZ, 0, range_for(ctZ * tl, wide), H, 0, 0, T);
      bld.ptr(d, 0, last(ws.DPHI16[0], xH), -xH); bld.ptr(d, 1, ws.fT[0]); bld.ptr(d, 2, nullptr); bld.ptr(d, 3, nullptr); bld.ptr(d, 4, nullptr);
      bld.ptr(d, 5, last(mu_q, sZ), -sZ); bld.ptr(d, 6, last(sd_q, sZ), -sZ); bld.ptr(d, 7, last(mu_p, sZ), -sZ); bld.ptr(d, 8, last(sd_p, sZ), -sZ);
      bld.ptr(d, 9, last(eps, sZ), -sZ); bld.ptr(d, 10, last(rs.RAWQ, sZ), -sZ); bld.ptr(d, 11, last(rs.RAWP, sZ), -sZ); bld.ptr(d, 12, nullptr);
      bld.ptr(d, 13, x_sl); bld.ptr(d, 14, c_raw); bld.ptr(d, 15, c_fn);
      bld.ptr(d, 16, last(ws.DQH, s2Z), -s2Z); bld.ptr(d, 17, last(ws.DQH16, x2Z), -x2Z); bld.ptr(d, 18, last(ws.DPH, s2Z), -s2Z);
      bld.ptr(d, 19, last(ws.DPH16, x2Z), -x2Z);
      d.ld[3] = 2 * Z; d.n16[0] = 2 * ctZ; d.i[0] = Z; d.i[1] = residual_posterior; d.i[2] = stride; d.i[3] = T - 1;
      d.f[0] = fn_floor; d.f[1] = beta; d.f[2] = sd_eps;
    }
    // B7: heads -> last hidden layers;  B8, B9: hidden layers 2, 1  (prior | posterior)
    const bool merge1 = vrnn_merge_first(groups);
    if (seq && merge1) {  // B7 .. B9 of a chain: one visit
      const SeqLinkB lp[3] = {{ws.phT, rs.P[2], ws.DP[2], ws.DP16[2]}, {ws.pT[2], rs.P[1], ws.DP[1], ws.DP16[1]}, {ws.pT[1], rs.P[0], ws.DP[0], ws.DP16[0]}};
      const SeqLinkB lq[3] = {{ws.qhT, rs.Q[2], ws.DQ[2], ws.DQ16[2]}, {ws.qT[2], rs.Q[1], ws.DQ[1], ws.DQ16[1]}, {ws.qT[1], rs.Q[0], ws.DQ[0], ws.DQ16[0]}};
      linseq_b(ws.DPH16, 3, lp, 0, half, 0, 2 * Z, x2Z);
      linseq_b(ws.DQH16, 3, lq, half, half, 0, 2 * Z, x2Z);
    }
    for (int l = 3; l >= (seq ? 3 : 1) && !(seq && merge1); --l) {
      lin(l == 3 ? ws.DPH16 : ws.DP16[l], l == 3 ? x2Z : xH, l == 3 ? ws.phT : ws.pT[l], l == 3 ? 2 * Z : H, nullptr, 0, 0, rs.P[l - 1], sH, H, ws.DP[l - 1], sH, H,
          ws.DP16[l - 1], xH, ctH, ctH, 0, half, 0);
      lin(l == 3 ? ws.DQH16 : ws.DQ16[l], l == 3 ? x2Z : xH, l == 3 ? ws.qhT : ws.qT[l], l == 3 ? 2 * Z : H, nullptr, 0, 0, rs.Q[l - 1], sH, H, ws.DQ[l - 1], sH, H,
          ws.DQ16[l - 1], xH, ctH, ctH, half, half, 0);
    }
    if (seq && !merge1) {  // B8, B9 of the prior | of the posterior: one visit each
      const SeqLinkB lp[2] = {{ws.pT[2], rs.P[1], ws.DP[1], ws.DP16[1]}, {ws.pT[1], rs.P[0], ws.DP[0], ws.DP16[0]}};
      const SeqLinkB lq[2] = {{ws.qT[2], rs.Q[1], ws.DQ[1], ws.DQ16[1]}, {ws.qT[1], rs.Q[0], ws.DQ[0], ws.DQ16[0]}};
      linseq_b(ws.DP16[2], 2, lp, 0, half, 0);
      linseq_b(ws.DQ16[2], 2, lq, half, half, 0);
    }
    BLVM_REQUIRE(!bld.overflow, "vrnn_bwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    // sentinel-fill what the launch polls: GA, GB (single words) and the T16 copies
    BLVM_HIP(pchain_fill_sentinel(ws.GA, (size_t)(reinterpret_cast<char*>(ws.x16_end) - reinterpret_cast<char*>(ws.GA)), s));
    // The batched GEMMs of the steps the chain has passed can run UNDER the rest of the chain (which keeps a quarter of the chip
    // busy): the sequence is cut into `parts` launches, after each the finished rows go to a low-priority side stream.
    const int parts = std::max(1, std::min(pchain_wgrad_parts(), T / 8));
    auto add_partials = [&](size_t r0, size_t nr) {  // DPHI[3] rows [r0, r0 + nr) += the two other partial sums
      if (!split3 || nr == 0) return;
      const size_t n4 = nr * H / 4;
      hipLaunchKernelGGL(add3_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s, reinterpret_cast<float4*>(ws.DPHI[3] + r0 * H),
                         reinterpret_cast<const float4*>(ws.DPHI3b + r0 * H), reinterpret_cast<const float4*>(ws.DPHI3c + r0 * H), n4);
    };
    if (parts == 1) {
      rc = pchain_launch(bld.p, s);
      if (rc) return rc;
      add_partials(0, n);
      return batched(0, n, s);
    }
    SideStream& sd = side_stream();
    rc = sd.ensure();
    if (rc) return rc;
    int done_steps = 0;  // steps s < done_steps are finished: rows of t in [T - done_steps, T)
    for (int k = 0; k < parts; ++k) {
      const int s_end = k + 1 == parts ? T + 1 : (int)((long)T * (k + 1) / parts);
      bld.p.s_first = done_steps; bld.p.S = s_end;
      if (k > 0) {  // a fresh abort epoch per launch
        rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
        if (rc) return rc;
      }
      rc = pchain_launch(bld.p, s);
      if (rc) return rc;
      const int fin = std::min(s_end, T);
      add_partials((size_t)(T - fin) * B, (size_t)(fin - done_steps) * B);
      // (all ranges on the ONE side stream: two of them accumulating into the same gradient must not run concurrently)
      BLVM_HIP(hipEventRecord(sd.ready, s));
      BLVM_HIP(hipStreamWaitEvent(sd.stream, sd.ready, 0));
      rc = batched((size_t)(T - fin) * B, (size_t)(fin - done_steps) * B, sd.stream);
      if (rc) return rc;
      done_steps = fin;
    }
    BLVM_HIP(hipEventRecord(sd.done, sd.stream));
    BLVM_HIP(hipStreamWaitEvent(s, sd.done, 0));
    return BLVM_OK;
  }
  const int chunk = overlap_chunk_steps();
  const bool overlap = chunk > 0 && Tp >= 2 * chunk;
  SideStream& side = side_stream();
  if (overlap) {
    rc = side.ensure();
    if (rc) return rc;
    // the side stream must not start before the caller's stream has produced this call's inputs / zeroed grads
    BLVM_HIP(hipEventRecord(side.ready, s));
    BLVM_HIP(hipStreamWaitEvent(side.stream, side.ready, 0));
  }
  int chunk_hi = Tp;  // steps [chunk_lo, chunk_hi) form the next range handed to the side stream

  launch_dh(-1, Tp - 1);  // G(T') = 0: gate derivatives of the last step, G <- d_decin h-part of row T'-1
  for (int t = Tp - 1; t >= 0; --t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z, o3R = (size_t)t * B * 3 * R, o2Z = (size_t)t * B * 2 * Z;
    const float* dec_t = decin + (size_t)t * B * ldd;
    const float* ddec_t = d_decin + (size_t)t * B * ldd;
    LinLaunch a;
    a.B = B;
    // B2: dphi (through ReLU of phi, plus the decoder's gradient) | G += DGH Whh
    a.nseg = 2;
    a.seg[0] = seg(ws.DGI + o3R, 3 * R, ws.wihT, 3 * R, nullptr, ddec_t, ldd, dec_t, ldd, ws.DPHI[3] + oH, H, H, 3 * R, 0);
    a.seg[1] = seg(ws.DGH + o3R, 3 * R, ws.whhT, 3 * R, nullptr, ws.G, R, nullptr, 0, ws.G, R, R, 3 * R, 0);
    launch_lin(a, s);
    // B3..B5: back through phi_z layers 3,2,1
    a.nseg = 1;
    for (int l = 3; l >= 1; --l) {
      a.seg[0] = seg(ws.DPHI[l] + oH, H, ws.fT[l], H, nullptr, nullptr, 0, rs.FZ[l - 1] + oH, H, ws.DPHI[l - 1] + oH, H, H, H, 0);
      launch_lin(a, s);
    }
    // B6: dz and the heads
    DzArgs d;
    d.D = ws.DPHI[0] + oH; d.WT = ws.fT[0]; d.D2 = nullptr; d.WT2 = nullptr; d.dz_add = nullptr; d.ld_add = 0; d.has_gemm = 1;
    d.mu_q = mu_q + oZ; d.sd_q = sd_q + oZ; d.mu_p = mu_p + oZ; d.sd_p = sd_p + oZ; d.eps = eps + oZ;
    d.raw_q = rs.RAWQ + oZ; d.raw_p = rs.RAWP + oZ;
    d.x_sl = x_sl; d.c_raw = c_raw; d.c_fn = c_fn;
    d.dqh = ws.DQH + o2Z; d.dph = ws.DPH + o2Z;
    d.B = B; d.H = H; d.Z = Z; d.residual = residual_posterior; d.t = t; d.stride = stride;
    d.fn_floor = fn_floor; d.beta = beta; d.sd_eps = sd_eps; d.muq_raw = nullptr;
    launch_dz(d, pick_nw(H, 1), dim3(Z / 16, rt), s);
    // B7: heads -> last hidden layers
    a.nseg = 2;
    a.seg[0] = seg(ws.DPH + o2Z, 2 * Z, ws.phT, 2 * Z, nullptr, nullptr, 0, rs.P[2] + oH, H, ws.DP[2] + oH, H, H, 2 * Z, 0);
    a.seg[1] = seg(ws.DQH + o2Z, 2 * Z, ws.qhT, 2 * Z, nullptr, nullptr, 0, rs.Q[2] + oH, H, ws.DQ[2] + oH, H, H, 2 * Z, 0);
    launch_lin(a, s);
    // B8, B9
    for (int l = 2; l >= 1; --l) {
      a.seg[0] = seg(ws.DP[l] + oH, H, ws.pT[l], H, nullptr, nullptr, 0, rs.P[l - 1] + oH, H, ws.DP[l - 1] + oH, H, H, H, 0);
      a.seg[1] = seg(ws.DQ[l] + oH, H, ws.qT[l], H, nullptr, nullptr, 0, rs.Q[l - 1] + oH, H, ws.DQ[l - 1] + oH, H, H, H, 0);
      launch_lin(a, s);
    }
    // B10 (+ gate derivatives of step t-1)
    launch_dh(t, t - 1);
    if (overlap && (chunk_hi - t >= chunk || t == 0)) {  // every per-step gradient of steps >= t is final now
      BLVM_HIP(hipEventRecord(side.ready, s));
      BLVM_HIP(hipStreamWaitEvent(side.stream, side.ready, 0));
      rc = batched((size_t)t * B, (size_t)(chunk_hi - t) * B, side.stream);
      if (rc) return rc;
      chunk_hi = t;
    }
  }
  BLVM_CHECK_LAUNCH("vrnn_seq_bwd");
  if (d_h0) BLVM_HIP(hipMemcpyAsync(d_h0, ws.G, sizeof(float) * (size_t)B * R, hipMemcpyDeviceToDevice, s));

  if (overlap) {  // join the side stream: everything queued after this call sees the weight gradients
    BLVM_HIP(hipEventRecord(side.done, side.stream));
    BLVM_HIP(hipStreamWaitEvent(s, side.done, 0));
  } else {
    rc = batched(0, n, s);
    if (rc) return rc;
  }
  return BLVM_OK;
}

extern "C" int blvm_vrnn_seq_bwd(const BlvmVrnnWeights* w, const float* enc, const float* eps, const float* decin,
                                 const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p,
                                 const float* z, const float* reserve, const float* d_decin, const int32_t* x_sl,
                                 const float* c_raw, const float* c_fn, int stride, float fn_floor, int Tp, int B, int X, int H, int Z,
                                 int R, int residual_posterior, float sd_eps, float* d_enc, float* d_h0,
                                 const BlvmVrnnGrads* gr, float* workspace, void* stream_) {
  BLVM_REQUIRE(w != nullptr && gr != nullptr, "vrnn_bwd: null pointer");
  auto body = [&](hipStream_t s) {
    return vrnn_seq_bwd_impl(w, enc, eps, decin, mu_q, sd_q, mu_p, sd_p, z, reserve, d_decin, x_sl, c_raw, c_fn, stride, fn_floor, Tp, B, X,
                             H, Z, R, residual_posterior, sd_eps, d_enc, d_h0, gr, workspace, s);
  };
  if (overlap_chunk_steps() > 0) return body(static_cast<hipStream_t>(stream_));  // the side-stream experiment is not capturable
  ChainKey key("vrnn_bwd");
  key.add(*w).add(enc).add(eps).add(decin).add(mu_q).add(sd_q).add(mu_p).add(sd_p).add(z).add(reserve).add(d_decin).add(x_sl);
  key.add(c_raw).add(c_fn).add(stride).add(fn_floor).add(Tp).add(B).add(X).add(H).add(Z).add(R).add(residual_posterior).add(sd_eps);
  key.add(d_enc).add(d_h0).add(*gr).add(workspace);
  return run_chain(key, static_cast<hipStream_t>(stream_), body);
}
