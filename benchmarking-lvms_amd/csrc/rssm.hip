// rssm.hip — K5: the recurrent state-space cell of the Clockwork-VAE over a whole sequence, forward + BPTT.
//
// Replaces the per-level time loop `clockwork_vae.py:272-281` over the scripted `RSSMCell.forward`
// (blvm/modules/rssm.py:79-104):
//   g = ReLU(Linear(cat[z_{t-1}, context_t]));  h_t = GRUCell(g, h_{t-1})
//   posterior = MLP(cat[h_t, enc_t]) -> (mu_q, sd_q);  prior = MLP(h_t) -> (mu_p, sd_p)
//   residual / precision-weighted combination (variational.py:125-138);  z_t = rsample
// Six dependent links per step in each direction (stages.h kernels + the two GRU links below); the context / encoding
// halves of the concatenated-input layers, every weight gradient and d(context), d(enc) are batched MFMA GEMMs
// outside the loop.  Same design and numerics as vrnn.hip.
#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {

#include "stages.h"

// ---- GRU link: gi = A Wih^T + bih (3 gate tiles) ; gates with the precomputed hidden projection ; state update ------
struct GruCellArgs {
  const float* A;      int lda;   // [B,K] GRU input
  const float* Wih;    int ldw;   // [3H,K] in T16 (ldw = K)
  const float* bih;               // [3H]
  const float* gh;                // [B,3H] hidden projection incl. b_hh
  const float* hprev;  int ldh;   // [B,H]
  float* hnext;        int ldn;   // [B,H]
  float *rg, *ug, *ng;            // [B,H]
  int B, H, K;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void gru_cell_stage_kernel(const float* A, const float* Wih, const float* bih,
                                                                 const float* gh, const float* hprev, unsigned lda_ldh,
                                                                 unsigned b_h, int K, float* hnext, int ldn, float* rg, float* ug,
                                                                 float* ng) {
  // scalar arguments (GruCellArgs documents them; ldw = K for a T16 weight): 5 input pointers, lda:16|ldh:16, B:16|H:16 and K are
  // the first 13 dwords, preloaded into SGPRs (stages.h lin1_stage_kernel)
  const int lda = lda_ldh & 0xffff, ldh = lda_ldh >> 16, B = b_h & 0xffff, H0 = b_h >> 16, ldw = K;
  __shared__ float red[3 * NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6, H = H0;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o3 = (size_t)rowc * 3 * H + col;
  const float b0 = bih[col], b1 = bih[H + col], b2 = bih[2 * H + col];
  const float hr = gh[o3], hz = gh[o3 + H], hn = gh[o3 + 2 * H];
  const float hp = hprev[(size_t)rowc * ldh + col];
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[3] = {A, A, A};
    const float* const Ws[3] = {Wih, Wih, Wih};
    const int la[3] = {lda, lda, lda}, lw[3] = {ldw, ldw, ldw}, cs[3] = {c0, H + c0, 2 * H + c0};
    wave_gemm16_multi<NW, 3, true>(As, la, r0, B, Ws, lw, cs, K, wave, acc);
  }
  float v[3];
  reduce_tiles<3, NW>(acc, red, v);
  if (!own) return;
  const float r = sigmoidf_(v[0] + b0 + hr);
  const float u = sigmoidf_(v[1] + b1 + hz);
  const float n = tanhf(v[2] + b2 + r * hn);
  hnext[(size_t)row * ldn + col] = (1.f - u) * n + u * hp;
  const size_t o = (size_t)row * H + col;
  rg[o] = r; ug[o] = u; ng[o] = n;
}

// ---- backward GRU link: complete dL/dh_t, then the gate derivatives of step t -------------------------------------------
struct RssmDhArgs {
  const float *DQ0, *DP0;   // [B,H] grads wrt the first posterior / prior layer pre-activations of step t
  const float *WqT, *WpT;   // [H,H] h-part of post_w0 / prior_w0, transposed, in T16
  const float* dh_add;      // [B,H] direct gradient wrt h_t (from the level below / the decoder) or null
  float* G;                 // [B,H] running gradient wrt h (in: from step t+1; out: towards step t-1 through the u gate)
  const float *rg, *ug, *ng, *gh;  // step t saves; gh [B,3H]
  const float* hprev;       // [B,H] h_{t-1}
  float *dgi, *dgh;         // [B,3H]
  int B, H;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void rssm_dh_stage_kernel(const float* DQ0, const float* DP0, const float* WqT,
                                                                const float* WpT, float* G, unsigned b_h, RssmDhArgs a) {
  // leading scalars preloaded into SGPRs, the struct by s_load, the saves prefetched in the `mid` hook (stages.h head_stage_kernel)
  __shared__ float red[2 * NW * 256];
  const int B = b_h & 0xffff, H = b_h >> 16;
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16, wave = threadIdx.x >> 6;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const bool own = threadIdx.x < 256 && row < B;
  const int rowc = row < B ? row : r0;
  const size_t o = (size_t)rowc * H + col, o3 = (size_t)rowc * 3 * H + col;
  const float gG = G[o];
  float gadd = 0.f, r = 0.f, u = 0.f, n = 0.f, hn = 0.f, hp = 0.f;
  auto prefetch = [&]() {
    gadd = a.dh_add != nullptr ? a.dh_add[o] : 0.f;
    r = a.rg[o]; u = a.ug[o]; n = a.ng[o]; hn = a.gh[o3 + 2 * H]; hp = a.hprev[o];
  };
  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* const As[2] = {DQ0, DP0};
    const float* const Ws[2] = {WqT, WpT};
    const int ld[2] = {H, H}, cs[2] = {c0, c0};
    wave_gemm16_multi<NW, 2, false>(As, ld, r0, B, Ws, ld, cs, H, wave, acc, prefetch);
  }
  float v[2];
  reduce_tiles<2, NW>(acc, red, v);
  if (!own) return;
  const float g0 = gG + gadd;
  const float g = g0 + v[0] + v[1];
  const float dn_pre = g * (1.f - u) * (1.f - n * n);
  const float du_pre = g * (hp - n) * u * (1.f - u);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  a.dgi[o3] = dr_pre; a.dgi[o3 + H] = du_pre; a.dgi[o3 + 2 * H] = dn_pre;
  a.dgh[o3] = dr_pre; a.dgh[o3 + H] = du_pre; a.dgh[o3 + 2 * H] = dn_pre * r;
  G[o] = g * u;
}

struct RssmReserve {
  float *GIN, *GHb, *RG, *UG, *NG, *Q[3], *P[3], *RAWQ, *RAWP, *MUQR, *XGIN, *XQ;
  float *Wgz, *Wih, *Whh, *Wq[3], *Wp[3], *Wqh, *Wph;  // T16 copies of the weights the forward chain multiplies by
  // persistent forward (B <= kPchainCarveMaxB): T16 copies of every activation a link multiplies, per step [rt*16, width]
  float *Z16, *H16, *GIN16, *Q16[3], *P16[3], *x16_end;
};
size_t carve_rssm(float* base, int T, int B, int H, int Z, RssmReserve* r) {
  const size_t n = (size_t)T * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  RssmReserve t;
  t.GIN = take(n * H); t.GHb = take(n * 3 * H);
  t.RG = take(n * H); t.UG = take(n * H); t.NG = take(n * H);
  for (int i = 0; i < 3; ++i) t.Q[i] = take(n * H);
  for (int i = 0; i < 3; ++i) t.P[i] = take(n * H);
  t.RAWQ = take(n * Z); t.RAWP = take(n * Z); t.MUQR = take(n * Z);
  t.XGIN = take(n * H); t.XQ = take(n * H);
  t.Wgz = take((size_t)H * Z); t.Wih = take((size_t)3 * H * H); t.Whh = take((size_t)3 * H * H);
  for (int i = 0; i < 3; ++i) { t.Wq[i] = take((size_t)H * H); t.Wp[i] = take((size_t)H * H); }
  t.Wqh = take((size_t)2 * Z * H); t.Wph = take((size_t)2 * Z * H);
  t.Z16 = nullptr;
  if (B <= kPchainCarveMaxB) {
    const size_t rows = (size_t)((B + 15) / 16) * 16, m = (size_t)T * rows;
    t.Z16 = take((m + rows) * Z); t.H16 = take((m + rows) * H); t.GIN16 = take(m * H);
    for (int i = 0; i < 3; ++i) { t.Q16[i] = take(m * H); t.P16[i] = take(m * H); }
    t.x16_end = take(0);
  }
  if (r) *r = t;
  return off;
}

struct RssmWs {
  float *gzT, *wihT, *whhT, *qT[3], *pT[3], *qhT, *phT, *DGIN, *DGI, *DGH, *DQH, *DPH, *DQ[3], *DP[3], *G;
  // persistent backward (B <= kPchainCarveMaxB): the running state gradient as per-step slabs [T,B,H] and T16 copies of every
  // gradient a link multiplies, per step [rt*16, width]
  float *GA, *GB, *DGIN16, *DGI16, *DGH16, *DQH16, *DPH16, *DQ16[3], *DP16[3], *x16_end;
};
size_t carve_rssm_ws(float* base, int T, int B, int H, int Z, RssmWs* w) {
  const size_t n = (size_t)T * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  RssmWs t;
  t.gzT = take((size_t)Z * H); t.wihT = take((size_t)H * 3 * H); t.whhT = take((size_t)H * 3 * H);
  for (int i = 0; i < 3; ++i) { t.qT[i] = take((size_t)H * H); t.pT[i] = take((size_t)H * H); }
  t.qhT = take((size_t)H * 2 * Z); t.phT = take((size_t)H * 2 * Z);
  t.DGIN = take(n * H); t.DGI = take(n * 3 * H); t.DGH = take(n * 3 * H);
  t.DQH = take(n * 2 * Z); t.DPH = take(n * 2 * Z);
  for (int i = 0; i < 3; ++i) { t.DQ[i] = take(n * H); t.DP[i] = take(n * H); }
  t.G = take((size_t)B * H);
  t.GA = nullptr;
  if (B <= kPchainCarveMaxB) {
    const size_t m = (size_t)T * ((B + 15) / 16) * 16;
    t.GA = take(n * H); t.GB = take(n * H);
    t.DGIN16 = take(m * H); t.DGI16 = take(m * 3 * H); t.DGH16 = take(m * 3 * H);
    t.DQH16 = take(m * 2 * Z); t.DPH16 = take(m * 2 * Z);
    for (int i = 0; i < 3; ++i) { t.DQ16[i] = take(m * H); t.DP16[i] = take(m * H); }
    t.x16_end = take(0);
  }
  if (w) *w = t;
  return off;
}

int check_rssm(int T, int B, int H, int Z, int C, int E) {
  BLVM_REQUIRE(T > 0 && B > 0, "rssm: bad T=%d B=%d", T, B);
  BLVM_REQUIRE(H > 0 && Z > 0 && H % 16 == 0 && Z % 16 == 0, "rssm: H, Z must be positive multiples of 16 (got %d, %d)", H, Z);
  BLVM_REQUIRE(C >= 0 && E > 0 && C % 4 == 0 && E % 4 == 0, "rssm: context / encoding sizes must be multiples of 4 (got %d, %d)", C, E);
  BLVM_REQUIRE(B < 65536 && H < 65536, "rssm: B and H must be below 65536 (packed kernel arguments)");
  return BLVM_OK;
}

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_rssm_reserve_floats(int T, int B, int H, int Z) { return carve_rssm(nullptr, T, B, H, Z, nullptr); }
extern "C" size_t blvm_rssm_bwd_workspace_floats(int T, int B, int H, int Z) { return carve_rssm_ws(nullptr, T, B, H, Z, nullptr); }

extern "C" int blvm_rssm_seq_fwd(const BlvmRssmWeights* w, const float* enc, const float* ctx, const float* z0,
                                 const float* h0, const float* eps, int T, int B, int H, int Z, int C, int E, int mode,
                                 float sd_eps, float* zs, float* hs, float* mu_q, float* sd_q, float* mu_p, float* sd_p,
                                 float* reserve, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rssm(T, B, H, Z, C, E);
  if (rc) return rc;
  BLVM_REQUIRE(w && enc && eps && zs && hs && mu_q && sd_q && mu_p && sd_p && reserve, "rssm_fwd: null pointer");
  BLVM_REQUIRE(C == 0 || ctx != nullptr, "rssm_fwd: context missing");
  BLVM_REQUIRE(mode >= 0 && mode <= 3, "rssm_fwd: mode must be 0 (plain), 1 (residual), 2 (precision-weighted) or 3 (generate: z from the prior)");
  BLVM_REQUIRE(aligned16(zs) && aligned16(hs) && aligned16(reserve), "rssm_fwd: buffers must be 16-byte aligned");
  RssmReserve rs;
  carve_rssm(reserve, T, B, H, Z, &rs);
  const size_t n = (size_t)T * B;
  const int ldg = Z + C, ldq = H + E;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
  if (C > 0) {
    rc = gemm_f32(0, 0, (int)n, H, C, ctx, C, w->gin_w + Z, ldg, rs.XGIN, H, w->gin_b, 0, 0.f, nullptr, 0, 0, 1, s);
    if (rc) return rc;
  }
  rc = gemm_f32(0, 0, (int)n, H, E, enc, E, w->post_w[0] + H, ldq, rs.XQ, H, w->post_b[0], 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  // T16 operand copies of the chain's weights (once per sequence): z columns of the GRU input layer, h columns of post_w0
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  rc = t16_pack_rows(w->gin_w, ldg, H, Z, rs.Wgz, s); if (rc) return rc;
  rc = t16_pack_rows(w->gru_wih, H, 3 * H, H, rs.Wih, s); if (rc) return rc;
  rc = t16_pack_rows(w->gru_whh, H, 3 * H, H, rs.Whh, s); if (rc) return rc;
  rc = t16_pack_rows(w->post_w[0], ldq, H, H, rs.Wq[0], s); if (rc) return rc;
  rc = t16_pack_rows(w->prior_w[0], H, H, H, rs.Wp[0], s); if (rc) return rc;
  for (int k = 1; k < 3; ++k) {
    rc = t16_pack_rows(w->post_w[k], H, H, H, rs.Wq[k], s); if (rc) return rc;
    rc = t16_pack_rows(w->prior_w[k], H, H, H, rs.Wp[k], s); if (rc) return rc;
  }
  rc = t16_pack_rows(w->post_hw, H, 2 * Z, H, rs.Wqh, s); if (rc) return rc;
  rc = t16_pack_rows(w->prior_hw, H, 2 * Z, H, rs.Wph, s); if (rc) return rc;
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;
  if (z0) BLVM_HIP(hipMemcpyAsync(zs, z0, sizeof(float) * (size_t)B * Z, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(zs, 0, sizeof(float) * (size_t)B * Z, s));
  if (h0) BLVM_HIP(hipMemcpyAsync(hs, h0, sizeof(float) * (size_t)B * H, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(hs, 0, sizeof(float) * (size_t)B * H, s));
  const int rt = (B + 15) / 16;
  if (pchain_applies(B) && device_cus() >= 32) {
    // Persistent path (pchain.h / pchain.hip): the six links of a step as a program of 10 descriptors, one launch per sequence.
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, cus = device_cus() & ~7;
    const long sH = (long)B * H, sZ = (long)B * Z, s3H = 3 * sH, xH = (long)rt * 16 * H, xZ = (long)rt * 16 * Z;
    const int r_h = range_for(ctH * rt, cus / 4);              // one H-wide link (or one half of a posterior | prior pair)
    const int r_gh = range_for(3 * ctH * rt, cus - 2 * r_h);   // the hidden projection, beside the GRU input layer
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.S = T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 4;
    bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = r_h;
    auto lin = [&](const float* A16, long a_step, const float* W, int K, const float* bias, const float* add, long add_step, float* orm, long rm_step,
                   int ldo, bool rm_sc1, float* o16, long o16_step, int ct, int wg0, int nwg, int flags) {
      Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags | (rm_sc1 ? DF_RM_SC1 : 0), 0, T);
      bld.ptr(d, 0, A16, a_step); bld.ptr(d, 1, W); bld.ptr(d, 2, bias); bld.ptr(d, 3, add, add_step); bld.ptr(d, 5, orm, rm_step);
      bld.ptr(d, 6, o16, o16_step);
      d.ld[1] = H; d.ld[3] = ldo; d.n16[0] = ctH; d.f[0] = 0.f;
    };
    // L1: GRU input layer (z half; the context half is hoisted) | hidden projection of the GRU
    lin(rs.Z16, xZ, rs.Wgz, Z, C > 0 ? nullptr : w->gin_b, C > 0 ? rs.XGIN : nullptr, sH, rs.GIN, sH, H, false, rs.GIN16, xH, ctH, 0, r_h, DF_RELU);
    lin(rs.H16, xH, rs.Whh, H, w->gru_bhh, nullptr, 0, rs.GHb, s3H, 3 * H, true, nullptr, 0, 3 * ctH, r_h, r_gh, 0);
    {  // L2: GRU
      Desc& d = bld.add(K_GRU, ctH, 0, r_h, H, 0, 0, T);
      bld.ptr(d, 0, rs.GIN16, xH); bld.ptr(d, 1, rs.Wih); bld.ptr(d, 2, nullptr); bld.ptr(d, 3, rs.GHb, s3H); bld.ptr(d, 4, hs, sH);
      bld.ptr(d, 5, hs + sH, sH); bld.ptr(d, 6, rs.H16 + xH, xH); bld.ptr(d, 7, rs.RG, sH); bld.ptr(d, 8, rs.UG, sH); bld.ptr(d, 9, rs.NG, sH);
      bld.ptr(d, 10, w->gru_bih);
      d.ld[0] = H; d.ld[3] = H; d.n16[0] = ctH; d.i[0] = H;
    }
    // L3..L5: posterior | prior MLPs on h_t
    const bool merge1 = linseq_enabled() && merge_first_enabled();  // (the first layer opens the run: one visit per branch and step)
    if (merge1) {
      const SeqLink lq[3] = {{rs.Wq[0], nullptr, rs.Q[0], sH, H, rs.Q16[0]}, {rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      const SeqLink lp[3] = {{rs.Wp[0], w->prior_b[0], rs.P[0], sH, H, rs.P16[0]}, {rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      add_linseq(bld, ctH, 0, r_h, H, true, false, 0, T, rs.H16 + xH, xH, 3, lq, 0, xH, ctH, 0.f, 0, 0, rs.XQ, sH, H);
      add_linseq(bld, ctH, r_h, r_h, H, true, false, 0, T, rs.H16 + xH, xH, 3, lp, 0, xH, ctH, 0.f, 0);
    } else {
    lin(rs.H16 + xH, xH, rs.Wq[0], H, nullptr, rs.XQ, sH, rs.Q[0], sH, H, false, rs.Q16[0], xH, ctH, 0, r_h, DF_RELU);
    lin(rs.H16 + xH, xH, rs.Wp[0], H, w->prior_b[0], nullptr, 0, rs.P[0], sH, H, false, rs.P16[0], xH, ctH, r_h, r_h, DF_RELU);
    }
    if (merge1) {
    } else if (linseq_enabled()) {  // layers 2, 3 of the posterior | prior MLP: one visit each (K_LINSEQ)
      const SeqLink lq[2] = {{rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      const SeqLink lp[2] = {{rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      add_linseq(bld, ctH, 0, r_h, H, true, false, 0, T, rs.Q16[0], xH, 2, lq, 0, xH, ctH, 0.f, 0);
      add_linseq(bld, ctH, r_h, r_h, H, true, false, 0, T, rs.P16[0], xH, 2, lp, 0, xH, ctH, 0.f, 0);
    } else {
      for (int k = 1; k < 3; ++k) {
        lin(rs.Q16[k - 1], xH, rs.Wq[k], H, w->post_b[k], nullptr, 0, rs.Q[k], sH, H, false, rs.Q16[k], xH, ctH, 0, r_h, DF_RELU);
        lin(rs.P16[k - 1], xH, rs.Wp[k], H, w->prior_b[k], nullptr, 0, rs.P[k], sH, H, false, rs.P16[k], xH, ctH, r_h, r_h, DF_RELU);
      }
    }
    {  // L6: heads, combination, sample
      Desc& d = bld.add(K_HEAD, ctZ, 0, range_for(ctZ * rt, 2 * r_h), H, 0, 0, T);
      bld.ptr(d, 0, rs.P16[2], xH); bld.ptr(d, 1, rs.Q16[2], xH); bld.ptr(d, 2, rs.Wph); bld.ptr(d, 3, w->prior_hb); bld.ptr(d, 4, rs.Wqh);
      bld.ptr(d, 5, w->post_hb); bld.ptr(d, 6, eps, sZ); bld.ptr(d, 7, mu_p, sZ); bld.ptr(d, 8, sd_p, sZ); bld.ptr(d, 9, mu_q, sZ);
      bld.ptr(d, 10, sd_q, sZ); bld.ptr(d, 11, rs.RAWP, sZ); bld.ptr(d, 12, rs.RAWQ, sZ); bld.ptr(d, 13, rs.MUQR, sZ);
      bld.ptr(d, 14, zs + sZ, sZ); bld.ptr(d, 15, rs.Z16 + xZ, xZ);
      d.ld[3] = Z; d.n16[0] = ctZ; d.i[0] = Z; d.i[1] = mode; d.f[0] = beta; d.f[1] = 1.f / beta; d.f[2] = sd_eps;
    }
    BLVM_REQUIRE(!bld.overflow, "rssm_fwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    // sentinel-fill what the launch polls: the T16 copies, the hidden projection, the states h_1 .. h_T (the GRU link polls words)
    BLVM_HIP(pchain_fill_sentinel(rs.Z16, (size_t)(reinterpret_cast<char*>(rs.x16_end) - reinterpret_cast<char*>(rs.Z16)), s));
    BLVM_HIP(pchain_fill_sentinel(rs.GHb, sizeof(float) * n * 3 * H, s));
    BLVM_HIP(pchain_fill_sentinel(hs + sH, sizeof(float) * n * H, s));
    rc = pchain_rows_to_t16(zs, Z, B, Z, rs.Z16, s); if (rc) return rc;
    rc = pchain_rows_to_t16(hs, H, B, H, rs.H16, s); if (rc) return rc;
    return pchain_launch(bld.p, s);
  }
  for (int t = 0; t < T; ++t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z, o3 = (size_t)t * B * 3 * H;
    const float* zprev = zs + oZ;
    const float* hprev = hs + oH;
    float* hnew = hs + oH + (size_t)B * H;
    LinLaunch l;
    l.B = B; l.nseg = 2;
    // L1: GRU input layer (z half; context half hoisted) | hidden projection of the GRU
    l.seg[0] = seg(zprev, Z, rs.Wgz, Z, C > 0 ? nullptr : w->gin_b, C > 0 ? rs.XGIN + oH : nullptr, H, nullptr, 0, rs.GIN + oH, H, H, Z, 1);
    l.seg[1] = seg(hprev, H, rs.Whh, H, w->gru_bhh, nullptr, 0, nullptr, 0, rs.GHb + o3, 3 * H, 3 * H, H, 0);
    launch_lin(l, s);
    // L2: GRU
    {
      const int nw = pick_nw(H, 3);
      const dim3 grid(H / 16, rt);
      const float *gin_t = rs.GIN + oH, *gh_t = rs.GHb + o3;
      float *rg_t = rs.RG + oH, *ug_t = rs.UG + oH, *ng_t = rs.NG + oH;
      const unsigned lda_ldh = (unsigned)H | ((unsigned)H << 16), b_h = (unsigned)B | ((unsigned)H << 16);
      if (nw == 16) hipLaunchKernelGGL((gru_cell_stage_kernel<16>), grid, dim3(1024), 0, s, gin_t, (const float*)rs.Wih, w->gru_bih, gh_t, hprev, lda_ldh, b_h, H, hnew, H, rg_t, ug_t, ng_t);
      else if (nw == 8) hipLaunchKernelGGL((gru_cell_stage_kernel<8>), grid, dim3(512), 0, s, gin_t, (const float*)rs.Wih, w->gru_bih, gh_t, hprev, lda_ldh, b_h, H, hnew, H, rg_t, ug_t, ng_t);
      else hipLaunchKernelGGL((gru_cell_stage_kernel<4>), grid, dim3(256), 0, s, gin_t, (const float*)rs.Wih, w->gru_bih, gh_t, hprev, lda_ldh, b_h, H, hnew, H, rg_t, ug_t, ng_t);
    }
    // L3..L5: posterior | prior MLPs on h_t
    l.seg[0] = seg(hnew, H, rs.Wq[0], H, nullptr, rs.XQ + oH, H, nullptr, 0, rs.Q[0] + oH, H, H, H, 1);
    l.seg[1] = seg(hnew, H, rs.Wp[0], H, w->prior_b[0], nullptr, 0, nullptr, 0, rs.P[0] + oH, H, H, H, 1);
    launch_lin(l, s);
    for (int k = 1; k < 3; ++k) {
      l.seg[0] = seg(rs.Q[k - 1] + oH, H, rs.Wq[k], H, w->post_b[k], nullptr, 0, nullptr, 0, rs.Q[k] + oH, H, H, H, 1);
      l.seg[1] = seg(rs.P[k - 1] + oH, H, rs.Wp[k], H, w->prior_b[k], nullptr, 0, nullptr, 0, rs.P[k] + oH, H, H, H, 1);
      launch_lin(l, s);
    }
    // L6: heads, combination, sample
    HeadArgs h;
    h.P = rs.P[2] + oH; h.Q = rs.Q[2] + oH;
    h.Wp = rs.Wph; h.bp = w->prior_hb; h.Wq = rs.Wqh; h.bq = w->post_hb;
    h.eps = eps + oZ;
    h.mu_p = mu_p + oZ; h.sd_p = sd_p + oZ; h.mu_q = mu_q + oZ; h.sd_q = sd_q + oZ;
    h.z = zs + oZ + (size_t)B * Z;
    h.raw_p = rs.RAWP + oZ; h.raw_q = rs.RAWQ + oZ; h.muq_raw = rs.MUQR + oZ;
    h.B = B; h.H = H; h.Z = Z; h.residual = mode;
    h.beta = beta; h.inv_beta = 1.f / beta; h.sd_eps = sd_eps;
    launch_head(h, pick_nw(H, 4), dim3(Z / 16, rt), s);
  }
  BLVM_CHECK_LAUNCH("rssm_seq_fwd");
  return BLVM_OK;
}

extern "C" int blvm_rssm_seq_bwd(const BlvmRssmWeights* w, const float* enc, const float* ctx, const float* eps,
                                 const float* zs, const float* hs, const float* mu_q, const float* sd_q,
                                 const float* mu_p, const float* sd_p, const float* reserve, const float* d_zs,
                                 const float* d_hs, const int32_t* x_sl, const float* c_raw, const float* c_fn, int stride,
                                 float fn_floor, int T, int B, int H, int Z, int C, int E, int mode, float sd_eps,
                                 float* d_enc, float* d_ctx, float* d_z0, float* d_h0, const BlvmRssmGrads* gr,
                                 float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_rssm(T, B, H, Z, C, E);
  if (rc) return rc;
  BLVM_REQUIRE(w && enc && eps && zs && hs && mu_q && sd_q && mu_p && sd_p && reserve && d_zs && d_hs && gr && workspace,
               "rssm_bwd: null pointer");
  BLVM_REQUIRE((c_fn == nullptr && c_raw == nullptr) || x_sl != nullptr, "rssm_bwd: KL coefficients need x_sl");
  BLVM_REQUIRE(aligned16(workspace) && aligned16(reserve), "rssm_bwd: buffers must be 16-byte aligned");
  RssmReserve rs;
  carve_rssm(const_cast<float*>(reserve), T, B, H, Z, &rs);
  RssmWs ws;
  carve_rssm_ws(workspace, T, B, H, Z, &ws);
  const size_t n = (size_t)T * B, bh = (size_t)B * H, bz = (size_t)B * Z;
  const int ldg = Z + C, ldq = H + E;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  TRY(t16_pack_transposed(w->gin_w, ldg, H, Z, ws.gzT, s));
  TRY(t16_pack_transposed(w->gru_wih, H, 3 * H, H, ws.wihT, s));
  TRY(t16_pack_transposed(w->gru_whh, H, 3 * H, H, ws.whhT, s));
  TRY(t16_pack_transposed(w->post_w[0], ldq, H, H, ws.qT[0], s));
  TRY(t16_pack_transposed(w->prior_w[0], H, H, H, ws.pT[0], s));
  for (int k = 1; k < 3; ++k) {
    TRY(t16_pack_transposed(w->post_w[k], H, H, H, ws.qT[k], s));
    TRY(t16_pack_transposed(w->prior_w[k], H, H, H, ws.pT[k], s));
  }
  TRY(t16_pack_transposed(w->post_hw, H, 2 * Z, H, ws.qhT, s));
  TRY(t16_pack_transposed(w->prior_hw, H, 2 * Z, H, ws.phT, s));
  TRY(pack_scope.flush());
  BLVM_HIP(hipMemsetAsync(ws.G, 0, sizeof(float) * bh, s));
  const int rt = (B + 15) / 16;
  const bool persistent = pchain_applies(B) && device_cus() >= 32;
  if (persistent) {
    // Persistent path: the BPTT chain as a program of 12 descriptors walked for s = 0 .. T (t = T-1-s; s = T: the gradients wrt the
    // initial state).  The running gradient wrt h lives in per-step slabs, each written once: GA[t] = g_t * u_t (GRU-backward link),
    // GB[t] = GA[t+1] + DGH[t+1] W_hh (a K = 3H product nothing needs before the GRU-backward link three links later: own range).
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, cus = device_cus() & ~7;
    const long sH = (long)B * H, sZ = (long)B * Z, s3H = 3 * sH, s2Z = 2 * sZ;
    const long xH = (long)rt * 16 * H, x3H = 3 * xH, x2Z = (long)rt * 16 * 2 * Z;
    const int r_h = range_for(ctH * rt, cus / 4), r_gb = range_for(ctH * rt, cus - 2 * r_h);
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.S = T + 1; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 2;
    bld.p.prof = pchain_profile_buffer() ? pchain_profile_buffer() + 64 : nullptr; bld.p.prof_wg = 2 * r_h;
    auto at = [&](const float* base, long step, int t0) { return base ? base + (long)t0 * step : nullptr; };  // slab of step t0
    auto lin = [&](const float* A16, long a_x, int a_t0, const float* W, int K, const float* add, long add_step, int add_t0, int ldadd, bool add_polled,
                   const float* gate, long gate_step, float* orm, long rm_step, int ldo, bool rm_sc1, float* o16, long o16_x, int n16, int ct, int wg0,
                   int nwg, int flags, int s0, int s1) {
      Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags | (add_polled ? DF_ADD_POLLED : 0) | (rm_sc1 ? DF_RM_SC1 : 0), s0, s1);
      bld.ptr(d, 0, at(A16, a_x, a_t0), -a_x); bld.ptr(d, 1, W); bld.ptr(d, 3, at(add, add_step, add_t0), -add_step);
      bld.ptr(d, 4, at(gate, gate_step, T - 1), -gate_step); bld.ptr(d, 5, rm_step ? at(orm, rm_step, T - 1) : orm, -rm_step);
      bld.ptr(d, 6, at(o16, o16_x, T - 1), -o16_x);
      d.ld[1] = ldadd; d.ld[2] = H; d.ld[3] = ldo; d.n16[0] = n16; d.f[0] = 0.f;
    };
    {  // B1: dz_t (direct + through the GRU input layer of step t+1), rsample / combination / KL / softplus heads
      Desc& d = bld.add(K_DZ, ctZ, 0, range_for(ctZ * rt, 2 * r_h), H, 0, 0, T);
      bld.ptr(d, 0, at(ws.DGIN16, xH, T), -xH); bld.ptr(d, 1, ws.gzT); bld.ptr(d, 4, at(d_zs, sZ, T), -sZ);
      bld.ptr(d, 5, at(mu_q, sZ, T - 1), -sZ); bld.ptr(d, 6, at(sd_q, sZ, T - 1), -sZ); bld.ptr(d, 7, at(mu_p, sZ, T - 1), -sZ);
      bld.ptr(d, 8, at(sd_p, sZ, T - 1), -sZ); bld.ptr(d, 9, at(eps, sZ, T - 1), -sZ); bld.ptr(d, 10, at(rs.RAWQ, sZ, T - 1), -sZ);
      bld.ptr(d, 11, at(rs.RAWP, sZ, T - 1), -sZ); bld.ptr(d, 12, at(rs.MUQR, sZ, T - 1), -sZ);
      bld.ptr(d, 13, x_sl); bld.ptr(d, 14, c_raw); bld.ptr(d, 15, c_fn);
      bld.ptr(d, 16, at(ws.DQH, s2Z, T - 1), -s2Z); bld.ptr(d, 17, at(ws.DQH16, x2Z, T - 1), -x2Z); bld.ptr(d, 18, at(ws.DPH, s2Z, T - 1), -s2Z);
      bld.ptr(d, 19, at(ws.DPH16, x2Z, T - 1), -x2Z);
      d.ld[1] = Z; d.ld[3] = 2 * Z; d.n16[0] = 2 * ctZ; d.i[0] = Z; d.i[1] = mode; d.i[2] = stride; d.i[3] = T - 1;
      d.f[0] = fn_floor; d.f[1] = beta; d.f[2] = sd_eps; d.f[3] = 1.f;
    }
    // B2: heads -> third layers (posterior | prior) | GB[t] = GA[t+1] + DGH[t+1] W_hh
    const bool merge1 = linseq_enabled() && merge_first_enabled();
    auto atm = [&](float* base, long step, int t0) { return base ? base + (long)t0 * step : nullptr; };
    if (merge1) {  // B2 .. B4 of a branch: one visit
      const SeqLink lq[3] = {{ws.qhT, at(rs.Q[2], sH, T - 1), atm(ws.DQ[2], sH, T - 1), -sH, H, atm(ws.DQ16[2], xH, T - 1)},
                             {ws.qT[2], at(rs.Q[1], sH, T - 1), atm(ws.DQ[1], sH, T - 1), -sH, H, atm(ws.DQ16[1], xH, T - 1)},
                             {ws.qT[1], at(rs.Q[0], sH, T - 1), atm(ws.DQ[0], sH, T - 1), -sH, H, atm(ws.DQ16[0], xH, T - 1)}};
      const SeqLink lp[3] = {{ws.phT, at(rs.P[2], sH, T - 1), atm(ws.DP[2], sH, T - 1), -sH, H, atm(ws.DP16[2], xH, T - 1)},
                             {ws.pT[2], at(rs.P[1], sH, T - 1), atm(ws.DP[1], sH, T - 1), -sH, H, atm(ws.DP16[1], xH, T - 1)},
                             {ws.pT[1], at(rs.P[0], sH, T - 1), atm(ws.DP[0], sH, T - 1), -sH, H, atm(ws.DP16[0], xH, T - 1)}};
      add_linseq(bld, ctH, 0, r_h, H, false, true, 0, T, at(ws.DQH16, x2Z, T - 1), -x2Z, 3, lq, -sH, -xH, ctH, 0.f, H, 2 * Z);
      add_linseq(bld, ctH, r_h, r_h, H, false, true, 0, T, at(ws.DPH16, x2Z, T - 1), -x2Z, 3, lp, -sH, -xH, ctH, 0.f, H, 2 * Z);
    } else {
    lin(ws.DQH16, x2Z, T - 1, ws.qhT, 2 * Z, nullptr, 0, 0, 0, false, rs.Q[2], sH, ws.DQ[2], sH, H, false, ws.DQ16[2], xH, ctH, ctH, 0, r_h, 0, 0, T);
    lin(ws.DPH16, x2Z, T - 1, ws.phT, 2 * Z, nullptr, 0, 0, 0, false, rs.P[2], sH, ws.DP[2], sH, H, false, ws.DP16[2], xH, ctH, ctH, r_h, r_h, 0, 0, T);
    }
    lin(ws.DGH16, x3H, T, ws.whhT, 3 * H, ws.GA, sH, T, H, true, nullptr, 0, ws.GB, sH, H, true, nullptr, 0, 0, ctH, 2 * r_h, r_gb, DF_GENTLE, 1, T);
    // B3, B4
    if (merge1) {
    } else if (linseq_enabled()) {  // one visit per branch (K_LINSEQ)
      const SeqLink lq[2] = {{ws.qT[2], at(rs.Q[1], sH, T - 1), atm(ws.DQ[1], sH, T - 1), -sH, H, atm(ws.DQ16[1], xH, T - 1)},
                             {ws.qT[1], at(rs.Q[0], sH, T - 1), atm(ws.DQ[0], sH, T - 1), -sH, H, atm(ws.DQ16[0], xH, T - 1)}};
      const SeqLink lp[2] = {{ws.pT[2], at(rs.P[1], sH, T - 1), atm(ws.DP[1], sH, T - 1), -sH, H, atm(ws.DP16[1], xH, T - 1)},
                             {ws.pT[1], at(rs.P[0], sH, T - 1), atm(ws.DP[0], sH, T - 1), -sH, H, atm(ws.DP16[0], xH, T - 1)}};
      add_linseq(bld, ctH, 0, r_h, H, false, true, 0, T, at(ws.DQ16[2], xH, T - 1), -xH, 2, lq, -sH, -xH, ctH, 0.f, H);
      add_linseq(bld, ctH, r_h, r_h, H, false, true, 0, T, at(ws.DP16[2], xH, T - 1), -xH, 2, lp, -sH, -xH, ctH, 0.f, H);
    } else {
      for (int k = 2; k >= 1; --k) {
        lin(ws.DQ16[k], xH, T - 1, ws.qT[k], H, nullptr, 0, 0, 0, false, rs.Q[k - 1], sH, ws.DQ[k - 1], sH, H, false, ws.DQ16[k - 1], xH, ctH, ctH, 0, r_h, 0, 0, T);
        lin(ws.DP16[k], xH, T - 1, ws.pT[k], H, nullptr, 0, 0, 0, false, rs.P[k - 1], sH, ws.DP[k - 1], sH, H, false, ws.DP16[k - 1], xH, ctH, ctH, r_h, r_h, 0, 0, T);
      }
    }
    {  // B5: complete dL/dh_t, gate derivatives of step t
      Desc& d = bld.add(K_GRUB, ctH, 0, r_h, H, 0, 0, T);
      bld.ptr(d, 0, at(ws.DQ16[0], xH, T - 1), -xH); bld.ptr(d, 1, at(ws.DP16[0], xH, T - 1), -xH); bld.ptr(d, 2, ws.qT[0]); bld.ptr(d, 3, ws.pT[0]);
      bld.ptr(d, 4, at(ws.GB, sH, T - 1), -sH);
      bld.ptr(d, 5, at(rs.RG, sH, T - 1), -sH); bld.ptr(d, 6, at(rs.UG, sH, T - 1), -sH); bld.ptr(d, 7, at(rs.NG, sH, T - 1), -sH);
      bld.ptr(d, 8, at(rs.GHb, s3H, T - 1), -s3H); bld.ptr(d, 9, at(hs, sH, T - 1), -sH); bld.ptr(d, 10, nullptr);
      bld.ptr(d, 11, at(ws.DGI, s3H, T - 1), -s3H); bld.ptr(d, 12, at(ws.DGI16, x3H, T - 1), -x3H); bld.ptr(d, 13, at(ws.DGH, s3H, T - 1), -s3H);
      bld.ptr(d, 14, at(ws.DGH16, x3H, T - 1), -x3H); bld.ptr(d, 15, at(ws.GA, sH, T - 1), -sH); bld.ptr(d, 16, ws.G);
      bld.ptr(d, 17, at(d_hs, sH, T), -sH);
      d.ld[0] = H; d.ld[1] = H; d.ld[3] = 3 * H; d.n16[0] = 3 * ctH; d.i[0] = H; d.i[1] = 0; d.i[2] = T; d.i[3] = 1;
    }
    // B6: through the GRU input projection to the (ReLU) GRU input layer
    lin(ws.DGI16, x3H, T - 1, ws.wihT, 3 * H, nullptr, 0, 0, 0, false, rs.GIN, sH, ws.DGIN, sH, H, false, ws.DGIN16, xH, ctH, ctH, 0, r_h, 0, 0, T);
    // s = T: gradients wrt the initial state: z0 through the GRU input layer of step 0 (+ its direct gradient), h0 through the GRU
    // of step 0 (the caller adds the direct gradient d_hs[0])
    if (d_z0) lin(ws.DGIN16, xH, T, ws.gzT, H, d_zs, sZ, T, Z, false, nullptr, 0, d_z0, 0, Z, false, nullptr, 0, 0, ctZ, 0, range_for(ctZ * rt, 2 * r_h), 0, T, T + 1);
    if (d_h0) lin(ws.DGH16, x3H, T, ws.whhT, 3 * H, ws.GA, sH, T, H, true, nullptr, 0, d_h0, 0, H, false, nullptr, 0, 0, ctH, 2 * r_h, r_gb, 0, T, T + 1);
    BLVM_REQUIRE(!bld.overflow, "rssm_bwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.GA, (size_t)(reinterpret_cast<char*>(ws.x16_end) - reinterpret_cast<char*>(ws.GA)), s));
    TRY(pchain_launch(bld.p, s));
  }
  for (int t = T - 1; t >= 0 && !persistent; --t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z, o3 = (size_t)t * B * 3 * H, o2Z = (size_t)t * B * 2 * Z;
    const bool last = t == T - 1;
    // B1: dz_t (direct + through the GRU input layer of step t+1), then rsample / combination / KL / softplus heads
    DzArgs dz;
    dz.has_gemm = !last;
    dz.D = ws.DGIN + (last ? 0 : oH + bh); dz.WT = ws.gzT; dz.D2 = nullptr; dz.WT2 = nullptr;
    dz.dz_add = d_zs + oZ + bz; dz.ld_add = Z;
    dz.mu_q = mu_q + oZ; dz.sd_q = sd_q + oZ; dz.mu_p = mu_p + oZ; dz.sd_p = sd_p + oZ; dz.eps = eps + oZ;
    dz.raw_q = rs.RAWQ + oZ; dz.raw_p = rs.RAWP + oZ; dz.muq_raw = rs.MUQR + oZ;
    dz.x_sl = x_sl; dz.c_raw = c_raw; dz.c_fn = c_fn;
    dz.dqh = ws.DQH + o2Z; dz.dph = ws.DPH + o2Z;
    dz.B = B; dz.H = H; dz.Z = Z; dz.residual = mode; dz.t = t; dz.stride = stride;
    dz.fn_floor = fn_floor; dz.beta = beta; dz.sd_eps = sd_eps;
    launch_dz(dz, pick_nw(H, 1), dim3(Z / 16, rt), s);
    // B2: heads -> third layers | G += DGH[t+1] Whh (the recurrent path of step t+1, off the critical chain)
    LinLaunch l;
    l.B = B; l.nseg = last ? 2 : 3;
    l.seg[0] = seg(ws.DQH + o2Z, 2 * Z, ws.qhT, 2 * Z, nullptr, nullptr, 0, rs.Q[2] + oH, H, ws.DQ[2] + oH, H, H, 2 * Z, 0);
    l.seg[1] = seg(ws.DPH + o2Z, 2 * Z, ws.phT, 2 * Z, nullptr, nullptr, 0, rs.P[2] + oH, H, ws.DP[2] + oH, H, H, 2 * Z, 0);
    if (!last) l.seg[2] = seg(ws.DGH + o3 + (size_t)B * 3 * H, 3 * H, ws.whhT, 3 * H, nullptr, ws.G, H, nullptr, 0, ws.G, H, H, 3 * H, 0);
    launch_lin(l, s);
    // B3, B4
    l.nseg = 2;
    for (int k = 2; k >= 1; --k) {
      l.seg[0] = seg(ws.DQ[k] + oH, H, ws.qT[k], H, nullptr, nullptr, 0, rs.Q[k - 1] + oH, H, ws.DQ[k - 1] + oH, H, H, H, 0);
      l.seg[1] = seg(ws.DP[k] + oH, H, ws.pT[k], H, nullptr, nullptr, 0, rs.P[k - 1] + oH, H, ws.DP[k - 1] + oH, H, H, H, 0);
      launch_lin(l, s);
    }
    // B5: complete dL/dh_t and the gate derivatives of step t
    RssmDhArgs d;
    d.DQ0 = ws.DQ[0] + oH; d.DP0 = ws.DP[0] + oH; d.WqT = ws.qT[0]; d.WpT = ws.pT[0];
    d.dh_add = d_hs + oH + bh; d.G = ws.G;
    d.rg = rs.RG + oH; d.ug = rs.UG + oH; d.ng = rs.NG + oH; d.gh = rs.GHb + o3; d.hprev = hs + oH;
    d.dgi = ws.DGI + o3; d.dgh = ws.DGH + o3; d.B = B; d.H = H;
    {
      const int nw = pick_nw(H, 2);
      const dim3 grid(H / 16, rt);
      const unsigned b_h = (unsigned)B | ((unsigned)H << 16);
      if (nw == 16) hipLaunchKernelGGL((rssm_dh_stage_kernel<16>), grid, dim3(1024), 0, s, d.DQ0, d.DP0, d.WqT, d.WpT, d.G, b_h, d);
      else if (nw == 8) hipLaunchKernelGGL((rssm_dh_stage_kernel<8>), grid, dim3(512), 0, s, d.DQ0, d.DP0, d.WqT, d.WpT, d.G, b_h, d);
      else hipLaunchKernelGGL((rssm_dh_stage_kernel<4>), grid, dim3(256), 0, s, d.DQ0, d.DP0, d.WqT, d.WpT, d.G, b_h, d);
    }
    // B6: through the GRU input projection to the (ReLU) GRU input layer
    l.nseg = 1;
    l.seg[0] = seg(ws.DGI + o3, 3 * H, ws.wihT, 3 * H, nullptr, nullptr, 0, rs.GIN + oH, H, ws.DGIN + oH, H, H, 3 * H, 0);
    launch_lin(l, s);
  }
  BLVM_CHECK_LAUNCH("rssm_seq_bwd");
  // gradients wrt the initial state: z0 through the GRU input layer of step 0 (+ its direct gradient), h0 through the
  // GRU of step 0 (the caller adds the direct gradient d_hs[0])
  if (d_z0 && !persistent) {
    LinLaunch l;
    l.B = B; l.nseg = 1;
    l.seg[0] = seg(ws.DGIN, H, ws.gzT, H, nullptr, d_zs, Z, nullptr, 0, d_z0, Z, Z, H, 0);
    launch_lin(l, s);
  }
  if (d_h0 && !persistent) {
    LinLaunch l;
    l.B = B; l.nseg = 1;
    l.seg[0] = seg(ws.DGH, 3 * H, ws.whhT, 3 * H, nullptr, ws.G, H, nullptr, 0, d_h0, H, H, 3 * H, 0);
    launch_lin(l, s);
  }
  BLVM_CHECK_LAUNCH("rssm_seq_bwd tail");
  // batched, state-independent part
  const float* hnew_all = hs + bh;  // h_1..h_T
  if (d_ctx && C > 0) TRY(gemm_f32(0, 1, (int)n, C, H, ws.DGIN, H, w->gin_w + Z, ldg, d_ctx, C, nullptr, 0, 0.f, nullptr, 0, 0, 1, s));
  if (d_enc) TRY(gemm_f32(0, 1, (int)n, E, H, ws.DQ[0], H, w->post_w[0] + H, ldq, d_enc, E, nullptr, 0, 0.f, nullptr, 0, 0, 1, s));
  WgradGroup grp;  // every weight gradient of the sequence: one grouped launch
  grp.add(ws.DGIN, H, H, zs, Z, Z, gr->gin_w, ldg, gr->gin_b);
  if (C > 0) grp.add(ws.DGIN, H, H, ctx, C, C, gr->gin_w ? gr->gin_w + Z : nullptr, ldg);
  grp.add(ws.DGI, 3 * H, 3 * H, rs.GIN, H, H, gr->gru_wih, H, gr->gru_bih);
  grp.add(ws.DGH, 3 * H, 3 * H, hs, H, H, gr->gru_whh, H, gr->gru_bhh);
  grp.add(ws.DQ[0], H, H, hnew_all, H, H, gr->post_w[0], ldq);
  grp.add(ws.DQ[0], H, H, enc, E, E, gr->post_w[0] ? gr->post_w[0] + H : nullptr, ldq, gr->post_b[0]);
  grp.add(ws.DP[0], H, H, hnew_all, H, H, gr->prior_w[0], H, gr->prior_b[0]);
  for (int k = 1; k < 3; ++k) {
    grp.add(ws.DQ[k], H, H, rs.Q[k - 1], H, H, gr->post_w[k], H, gr->post_b[k]);
    grp.add(ws.DP[k], H, H, rs.P[k - 1], H, H, gr->prior_w[k], H, gr->prior_b[k]);
  }
  grp.add(ws.DQH, 2 * Z, 2 * Z, rs.Q[2], H, H, gr->post_hw, H, gr->post_hb);
  grp.add(ws.DPH, 2 * Z, 2 * Z, rs.P[2], H, H, gr->prior_hw, H, gr->prior_hb);
  TRY(grp.run(n, s));
#undef TRY
  return BLVM_OK;
}
