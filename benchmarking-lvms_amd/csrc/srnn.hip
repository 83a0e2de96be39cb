// srnn.hip — K3: the SRNN latent chain (forward + BPTT) as a stage-kernel chain.
//
// Replaces the Python loop over time of the reference (blvm/models/srnn.py:224-253): per step
//   h_p = cat[d_t, z_{t-1}], h_q = cat[a_t, z_{t-1}];  prior / posterior = 3 x (Linear + LeakyReLU) + DiagonalGaussianDense
//   (srnn.py:92-111);  enc_mu += prior_mu (residual posterior);  z_t = rsample
// and its autograd backward.  Only the z-halves of the two first layers are recurrent: the d/a halves are hoisted
// into one MFMA GEMM each before the loop, so a step is 4 dependent links (first layers | second | third | heads +
// sample) forward and 4 backward, every element-wise piece fused into an epilogue (stages.h).  The KL(+free nats)
// gradient is folded into the backward chain exactly as for the VRNN cell.
#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {

#include "stages.h"

struct SrnnReserve {
  float *P[3], *Q[3], *XP, *XQ, *RAWP, *RAWQ, *Wp[3], *Wq[3], *Wph, *Wqh;  // W*: T16 weight copies
  float *Z16, *P16[3], *Q16[3], *x16_end;  // persistent forward (B <= kPchainCarveMaxB): T16 copies of what the links multiply
};
size_t carve_srnn(float* base, int Tp, int B, int H, int Z, SrnnReserve* r) {
  const size_t n = (size_t)Tp * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  SrnnReserve t;
  for (int i = 0; i < 3; ++i) t.P[i] = take(n * H);
  for (int i = 0; i < 3; ++i) t.Q[i] = take(n * H);
  t.XP = take(n * H); t.XQ = take(n * H);
  t.RAWP = take(n * Z); t.RAWQ = take(n * Z);
  t.Wp[0] = take((size_t)H * Z); t.Wq[0] = take((size_t)H * Z);
  for (int i = 1; i < 3; ++i) { t.Wp[i] = take((size_t)H * H); t.Wq[i] = take((size_t)H * H); }
  t.Wph = take((size_t)2 * Z * H); t.Wqh = take((size_t)2 * Z * H);
  t.Z16 = nullptr;
  if (B <= kPchainCarveMaxB) {
    const size_t rows = (size_t)((B + 15) / 16) * 16, m = (size_t)Tp * rows;
    t.Z16 = take((m + rows) * Z);
    for (int i = 0; i < 3; ++i) { t.P16[i] = take(m * H); t.Q16[i] = take(m * H); }
    t.x16_end = take(0);
  }
  if (r) *r = t;
  return off;
}

struct SrnnWs {
  float *pzT, *qzT, *pT[3], *qT[3], *phT, *qhT, *DPH, *DQH, *DP[3], *DQ[3];
  float *DZ0, *DPH16, *DQH16, *DP16[3], *DQ16[3], *x16_end;  // persistent backward (B <= kPchainCarveMaxB)
};
size_t carve_srnn_ws(float* base, int Tp, int B, int H, int Z, SrnnWs* w) {
  const size_t n = (size_t)Tp * B;
  size_t off = 0;
  auto take = [&](size_t cnt) { float* p = base ? base + off : nullptr; off += (cnt + 3) & ~(size_t)3; return p; };
  SrnnWs t;
  t.pzT = take((size_t)Z * H); t.qzT = take((size_t)Z * H);
  t.pT[0] = t.qT[0] = nullptr;
  for (int i = 1; i < 3; ++i) { t.pT[i] = take((size_t)H * H); t.qT[i] = take((size_t)H * H); }
  t.phT = take((size_t)H * 2 * Z); t.qhT = take((size_t)H * 2 * Z);
  t.DPH = take(n * 2 * Z); t.DQH = take(n * 2 * Z);
  for (int i = 0; i < 3; ++i) { t.DP[i] = take(n * H); t.DQ[i] = take(n * H); }
  t.DZ0 = nullptr;
  if (B <= kPchainCarveMaxB) {
    const size_t m = (size_t)Tp * ((B + 15) / 16) * 16;
    t.DZ0 = take((size_t)B * Z);
    t.DPH16 = take(m * 2 * Z); t.DQH16 = take(m * 2 * Z);
    for (int i = 0; i < 3; ++i) { t.DP16[i] = take(m * H); t.DQ16[i] = take(m * H); }
    t.x16_end = take(0);
  }
  if (w) *w = t;
  return off;
}

int check_srnn(int Tp, int B, int H, int Z, int R) {
  BLVM_REQUIRE(Tp > 0 && B > 0, "srnn: bad Tp=%d B=%d", Tp, B);
  BLVM_REQUIRE(H > 0 && Z > 0 && R > 0 && H % 16 == 0 && Z % 16 == 0 && R % 16 == 0,
               "srnn: H,Z,R must be positive multiples of 16 (got %d,%d,%d)", H, Z, R);
  BLVM_REQUIRE(B < 65536 && H < 65536, "srnn: B and H must be below 65536 (packed kernel arguments)");
  return BLVM_OK;
}

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_srnn_reserve_floats(int Tp, int B, int H, int Z, int R) {
  (void)R;
  return carve_srnn(nullptr, Tp, B, H, Z, nullptr);
}
extern "C" size_t blvm_srnn_bwd_workspace_floats(int Tp, int B, int H, int Z, int R) {
  (void)R;
  return carve_srnn_ws(nullptr, Tp, B, H, Z, nullptr);
}

extern "C" int blvm_srnn_latent_fwd(const BlvmSrnnWeights* w, const float* d, const float* a, const float* z0,
                                    const float* eps, int Tp, int B, int H, int Z, int R, int residual_posterior,
                                    float sd_eps, float slope, float* zs, float* mu_q, float* sd_q, float* mu_p,
                                    float* sd_p, float* reserve, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_srnn(Tp, B, H, Z, R);
  if (rc) return rc;
  BLVM_REQUIRE(w && d && a && eps && zs && mu_q && sd_q && mu_p && sd_p && reserve, "srnn_fwd: null pointer");
  BLVM_REQUIRE(aligned16(zs) && aligned16(reserve) && aligned16(d) && aligned16(a), "srnn_fwd: buffers must be 16-byte aligned");
  SrnnReserve rs;
  carve_srnn(reserve, Tp, B, H, Z, &rs);
  const size_t n = (size_t)Tp * B;
  const int ldw0 = R + Z;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
  // hoisted d / a halves of the two first layers (incl. bias)
  rc = gemm_f32(0, 0, (int)n, H, R, d, R, w->prior_w[0], ldw0, rs.XP, H, w->prior_b[0], 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  rc = gemm_f32(0, 0, (int)n, H, R, a, R, w->post_w[0], ldw0, rs.XQ, H, w->post_b[0], 0, 0.f, nullptr, 0, 0, 1, s);
  if (rc) return rc;
  // T16 operand copies of the chain's weights (once per sequence); layer 0: the z columns
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  rc = t16_pack_rows(w->prior_w[0] + R, ldw0, H, Z, rs.Wp[0], s); if (rc) return rc;
  rc = t16_pack_rows(w->post_w[0] + R, ldw0, H, Z, rs.Wq[0], s); if (rc) return rc;
  for (int k = 1; k < 3; ++k) {
    rc = t16_pack_rows(w->prior_w[k], H, H, H, rs.Wp[k], s); if (rc) return rc;
    rc = t16_pack_rows(w->post_w[k], H, H, H, rs.Wq[k], s); if (rc) return rc;
  }
  rc = t16_pack_rows(w->prior_hw, H, 2 * Z, H, rs.Wph, s); if (rc) return rc;
  rc = t16_pack_rows(w->post_hw, H, 2 * Z, H, rs.Wqh, s); if (rc) return rc;
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;
  if (z0) BLVM_HIP(hipMemcpyAsync(zs, z0, sizeof(float) * (size_t)B * Z, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(zs, 0, sizeof(float) * (size_t)B * Z, s));
  const int rt = (B + 15) / 16;
  if (pchain_applies(B) && device_cus() >= 32) {
    // Persistent path (pchain.h / pchain.hip): the four links of a step as a program of 7 descriptors, one launch per sequence
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, cus = device_cus() & ~7;
    const long sH = (long)B * H, sZ = (long)B * Z, xH = (long)rt * 16 * H, xZ = (long)rt * 16 * Z;
    const int half = range_for(ctH * rt, cus / 2);
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.S = Tp; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 4;
    bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = half;
    auto lin = [&](const float* A16, long a_step, const float* W, int K, const float* bias, const float* add, float* orm, float* o16, int wg0) {
      Desc& d = bld.add(K_LIN, ctH, wg0, half, K, DF_RELU, 0, Tp);
      bld.ptr(d, 0, A16, a_step); bld.ptr(d, 1, W); bld.ptr(d, 2, bias); bld.ptr(d, 3, add, sH); bld.ptr(d, 5, orm, sH); bld.ptr(d, 6, o16, xH);
      d.ld[1] = H; d.ld[3] = H; d.n16[0] = ctH; d.f[0] = slope;
    };
    const bool merge1 = linseq_enabled() && merge_first_enabled();  // (the first layer opens the run: one visit per chain and step)
    if (merge1) {
      const SeqLink lp[3] = {{rs.Wp[0], nullptr, rs.P[0], sH, H, rs.P16[0]}, {rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      const SeqLink lq[3] = {{rs.Wq[0], nullptr, rs.Q[0], sH, H, rs.Q16[0]}, {rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      add_linseq(bld, ctH, 0, half, H, true, false, 0, Tp, rs.Z16, xZ, 3, lp, 0, xH, ctH, slope, 0, Z, rs.XP, sH, H);
      add_linseq(bld, ctH, half, half, H, true, false, 0, Tp, rs.Z16, xZ, 3, lq, 0, xH, ctH, slope, 0, Z, rs.XQ, sH, H);
    } else {
    lin(rs.Z16, xZ, rs.Wp[0], Z, nullptr, rs.XP, rs.P[0], rs.P16[0], 0);
    lin(rs.Z16, xZ, rs.Wq[0], Z, nullptr, rs.XQ, rs.Q[0], rs.Q16[0], half);
    }
    if (merge1) {
    } else if (linseq_enabled()) {  // layers 2, 3 of the prior | posterior MLP: one visit each
      const SeqLink lp[2] = {{rs.Wp[1], w->prior_b[1], rs.P[1], sH, H, rs.P16[1]}, {rs.Wp[2], w->prior_b[2], rs.P[2], sH, H, rs.P16[2]}};
      const SeqLink lq[2] = {{rs.Wq[1], w->post_b[1], rs.Q[1], sH, H, rs.Q16[1]}, {rs.Wq[2], w->post_b[2], rs.Q[2], sH, H, rs.Q16[2]}};
      add_linseq(bld, ctH, 0, half, H, true, false, 0, Tp, rs.P16[0], xH, 2, lp, 0, xH, ctH, slope, 0);
      add_linseq(bld, ctH, half, half, H, true, false, 0, Tp, rs.Q16[0], xH, 2, lq, 0, xH, ctH, slope, 0);
    } else {
      for (int k = 1; k < 3; ++k) {
        lin(rs.P16[k - 1], xH, rs.Wp[k], H, w->prior_b[k], nullptr, rs.P[k], rs.P16[k], 0);
        lin(rs.Q16[k - 1], xH, rs.Wq[k], H, w->post_b[k], nullptr, rs.Q[k], rs.Q16[k], half);
      }
    }
    {
      Desc& d = bld.add(K_HEAD, ctZ, 0, range_for(ctZ * rt, 2 * half), H, 0, 0, Tp);
      bld.ptr(d, 0, rs.P16[2], xH); bld.ptr(d, 1, rs.Q16[2], xH); bld.ptr(d, 2, rs.Wph); bld.ptr(d, 3, w->prior_hb); bld.ptr(d, 4, rs.Wqh);
      bld.ptr(d, 5, w->post_hb); bld.ptr(d, 6, eps, sZ); bld.ptr(d, 7, mu_p, sZ); bld.ptr(d, 8, sd_p, sZ); bld.ptr(d, 9, mu_q, sZ);
      bld.ptr(d, 10, sd_q, sZ); bld.ptr(d, 11, rs.RAWP, sZ); bld.ptr(d, 12, rs.RAWQ, sZ); bld.ptr(d, 13, nullptr);
      bld.ptr(d, 14, zs + sZ, sZ); bld.ptr(d, 15, rs.Z16 + xZ, xZ);
      d.ld[3] = Z; d.n16[0] = ctZ; d.i[0] = Z; d.i[1] = residual_posterior; d.f[0] = beta; d.f[1] = 1.f / beta; d.f[2] = sd_eps;
    }
    BLVM_REQUIRE(!bld.overflow, "srnn_fwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(rs.Z16, (size_t)(reinterpret_cast<char*>(rs.x16_end) - reinterpret_cast<char*>(rs.Z16)), s));
    rc = pchain_rows_to_t16(zs, Z, B, Z, rs.Z16, s);
    if (rc) return rc;
    return pchain_launch(bld.p, s);
  }
  for (int t = 0; t < Tp; ++t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z;
    const float* zprev = zs + oZ;
    LinLaunch l;
    l.B = B; l.slope = slope; l.nseg = 2;
    l.seg[0] = seg(zprev, Z, rs.Wp[0], Z, nullptr, rs.XP + oH, H, nullptr, 0, rs.P[0] + oH, H, H, Z, 1);
    l.seg[1] = seg(zprev, Z, rs.Wq[0], Z, nullptr, rs.XQ + oH, H, nullptr, 0, rs.Q[0] + oH, H, H, Z, 1);
    launch_lin(l, s);
    for (int k = 1; k < 3; ++k) {
      l.seg[0] = seg(rs.P[k - 1] + oH, H, rs.Wp[k], H, w->prior_b[k], nullptr, 0, nullptr, 0, rs.P[k] + oH, H, H, H, 1);
      l.seg[1] = seg(rs.Q[k - 1] + oH, H, rs.Wq[k], H, w->post_b[k], nullptr, 0, nullptr, 0, rs.Q[k] + oH, H, H, H, 1);
      launch_lin(l, s);
    }
    HeadArgs h;
    h.P = rs.P[2] + oH; h.Q = rs.Q[2] + oH;
    h.Wp = rs.Wph; h.bp = w->prior_hb; h.Wq = rs.Wqh; h.bq = w->post_hb;
    h.eps = eps + oZ;
    h.mu_p = mu_p + oZ; h.sd_p = sd_p + oZ; h.mu_q = mu_q + oZ; h.sd_q = sd_q + oZ;
    h.z = zs + (size_t)(t + 1) * B * Z;
    h.raw_p = rs.RAWP + oZ; h.raw_q = rs.RAWQ + oZ;
    h.B = B; h.H = H; h.Z = Z; h.residual = residual_posterior;
    h.beta = beta; h.inv_beta = 1.f / beta; h.sd_eps = sd_eps; h.muq_raw = nullptr;
    launch_head(h, pick_nw(H, 4), dim3(Z / 16, rt), s);
  }
  BLVM_CHECK_LAUNCH("srnn_latent_fwd");
  return BLVM_OK;
}

extern "C" int blvm_srnn_latent_bwd(const BlvmSrnnWeights* w, const float* d, const float* a, const float* eps,
                                    const float* zs, const float* mu_q, const float* sd_q, const float* mu_p,
                                    const float* sd_p, const float* reserve, const float* d_z, const int32_t* x_sl,
                                    const float* c_raw, const float* c_fn, int stride, float fn_floor, int Tp, int B,
                                    int H, int Z, int R, int residual_posterior, float sd_eps, float slope, float* d_d,
                                    float* d_a, float* d_z0, const BlvmSrnnGrads* gr, float* workspace, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  int rc = check_srnn(Tp, B, H, Z, R);
  if (rc) return rc;
  BLVM_REQUIRE(w && d && a && eps && zs && mu_q && sd_q && mu_p && sd_p && reserve && d_z && workspace && gr, "srnn_bwd: null pointer");
  BLVM_REQUIRE((c_fn == nullptr && c_raw == nullptr) || x_sl != nullptr, "srnn_bwd: KL coefficients need x_sl");
  BLVM_REQUIRE(aligned16(workspace) && aligned16(reserve), "srnn_bwd: buffers must be 16-byte aligned");
  SrnnReserve rs;
  carve_srnn(const_cast<float*>(reserve), Tp, B, H, Z, &rs);
  SrnnWs ws;
  carve_srnn_ws(workspace, Tp, B, H, Z, &ws);
  const size_t n = (size_t)Tp * B;
  const int ldw0 = R + Z;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
  T16PackScope pack_scope(pchain_bf16(B), s);  // bf16-operand mode: the persistent launch multiplies bf16 weight packs
  TRY(t16_pack_transposed(w->prior_w[0] + R, ldw0, H, Z, ws.pzT, s));
  TRY(t16_pack_transposed(w->post_w[0] + R, ldw0, H, Z, ws.qzT, s));
  for (int k = 1; k < 3; ++k) {
    TRY(t16_pack_transposed(w->prior_w[k], H, H, H, ws.pT[k], s));
    TRY(t16_pack_transposed(w->post_w[k], H, H, H, ws.qT[k], s));
  }
  TRY(t16_pack_transposed(w->prior_hw, H, 2 * Z, H, ws.phT, s));
  TRY(t16_pack_transposed(w->post_hw, H, 2 * Z, H, ws.qhT, s));
  TRY(pack_scope.flush());
  const int rt = (B + 15) / 16;
  const bool persistent = pchain_applies(B) && device_cus() >= 32;
  if (persistent) {
    // Persistent path: the BPTT chain as a program of 9 descriptors walked for s = 0 .. T' (t = T'-1-s; s = T': the gradient wrt z_0)
    using namespace pchain;
    const int ctH = H / 16, ctZ = Z / 16, cus = device_cus() & ~7, T = Tp;
    const long sH = (long)B * H, sZ = (long)B * Z, s2Z = 2 * sZ, xH = (long)rt * 16 * H, x2Z = (long)rt * 16 * 2 * Z;
    const int half = range_for(ctH * rt, cus / 2);
    Builder bld;
    bld.p.bf16 = pchain_bf16(B);
    bld.p.S = T + 1; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 2;
    bld.p.prof = pchain_profile_buffer() ? pchain_profile_buffer() + 64 : nullptr; bld.p.prof_wg = half;
    auto at = [&](const float* base, long step, int t0) { return base ? base + (long)t0 * step : nullptr; };
    {  // B1: dz_t = decoder gradient + the two first layers of step t+1, then rsample / residual / KL / softplus heads
      Desc& d = bld.add(K_DZ, ctZ, 0, range_for(ctZ * rt, 2 * half), H, 0, 0, T);
      bld.ptr(d, 0, at(ws.DP16[0], xH, T), -xH); bld.ptr(d, 1, ws.pzT); bld.ptr(d, 2, at(ws.DQ16[0], xH, T), -xH); bld.ptr(d, 3, ws.qzT);
      bld.ptr(d, 4, at(d_z, sZ, T - 1), -sZ);
      bld.ptr(d, 5, at(mu_q, sZ, T - 1), -sZ); bld.ptr(d, 6, at(sd_q, sZ, T - 1), -sZ); bld.ptr(d, 7, at(mu_p, sZ, T - 1), -sZ);
      bld.ptr(d, 8, at(sd_p, sZ, T - 1), -sZ); bld.ptr(d, 9, at(eps, sZ, T - 1), -sZ); bld.ptr(d, 10, at(rs.RAWQ, sZ, T - 1), -sZ);
      bld.ptr(d, 11, at(rs.RAWP, sZ, T - 1), -sZ); bld.ptr(d, 12, nullptr); bld.ptr(d, 13, x_sl); bld.ptr(d, 14, c_raw); bld.ptr(d, 15, c_fn);
      bld.ptr(d, 16, at(ws.DQH, s2Z, T - 1), -s2Z); bld.ptr(d, 17, at(ws.DQH16, x2Z, T - 1), -x2Z); bld.ptr(d, 18, at(ws.DPH, s2Z, T - 1), -s2Z);
      bld.ptr(d, 19, at(ws.DPH16, x2Z, T - 1), -x2Z);
      d.ld[1] = Z; d.ld[3] = 2 * Z; d.n16[0] = 2 * ctZ; d.i[0] = Z; d.i[1] = residual_posterior; d.i[2] = stride; d.i[3] = T - 1;
      d.f[0] = fn_floor; d.f[1] = beta; d.f[2] = sd_eps; d.f[3] = 1.f;
    }
    auto lin = [&](const float* A16, long a_x, const float* W, int K, const float* gate, float* orm, float* o16, int wg0) {
      Desc& d = bld.add(K_LIN, ctH, wg0, half, K, 0, 0, T);
      bld.ptr(d, 0, at(A16, a_x, T - 1), -a_x); bld.ptr(d, 1, W); bld.ptr(d, 4, at(gate, sH, T - 1), -sH); bld.ptr(d, 5, at(orm, sH, T - 1), -sH);
      bld.ptr(d, 6, at(o16, xH, T - 1), -xH);
      d.ld[2] = H; d.ld[3] = H; d.n16[0] = ctH; d.f[0] = slope;
    };
    // B2: heads -> third layers;  B3, B4: down to the first layers (LeakyReLU derivatives fused)
    const bool merge1 = linseq_enabled() && merge_first_enabled();
    auto atm = [&](float* base, long step, int t0) { return base ? base + (long)t0 * step : nullptr; };
    if (merge1) {  // B2 .. B4 of a chain: one visit
      const SeqLink lp[3] = {{ws.phT, at(rs.P[2], sH, T - 1), atm(ws.DP[2], sH, T - 1), -sH, H, atm(ws.DP16[2], xH, T - 1)},
                             {ws.pT[2], at(rs.P[1], sH, T - 1), atm(ws.DP[1], sH, T - 1), -sH, H, atm(ws.DP16[1], xH, T - 1)},
                             {ws.pT[1], at(rs.P[0], sH, T - 1), atm(ws.DP[0], sH, T - 1), -sH, H, atm(ws.DP16[0], xH, T - 1)}};
      const SeqLink lq[3] = {{ws.qhT, at(rs.Q[2], sH, T - 1), atm(ws.DQ[2], sH, T - 1), -sH, H, atm(ws.DQ16[2], xH, T - 1)},
                             {ws.qT[2], at(rs.Q[1], sH, T - 1), atm(ws.DQ[1], sH, T - 1), -sH, H, atm(ws.DQ16[1], xH, T - 1)},
                             {ws.qT[1], at(rs.Q[0], sH, T - 1), atm(ws.DQ[0], sH, T - 1), -sH, H, atm(ws.DQ16[0], xH, T - 1)}};
      add_linseq(bld, ctH, 0, half, H, false, true, 0, T, at(ws.DPH16, x2Z, T - 1), -x2Z, 3, lp, -sH, -xH, ctH, slope, H, 2 * Z);
      add_linseq(bld, ctH, half, half, H, false, true, 0, T, at(ws.DQH16, x2Z, T - 1), -x2Z, 3, lq, -sH, -xH, ctH, slope, H, 2 * Z);
    } else {
    lin(ws.DPH16, x2Z, ws.phT, 2 * Z, rs.P[2], ws.DP[2], ws.DP16[2], 0);
    lin(ws.DQH16, x2Z, ws.qhT, 2 * Z, rs.Q[2], ws.DQ[2], ws.DQ16[2], half);
    }
    if (merge1) {
    } else if (linseq_enabled()) {  // B3, B4 of the prior | posterior: one visit each
      const SeqLink lp[2] = {{ws.pT[2], at(rs.P[1], sH, T - 1), atm(ws.DP[1], sH, T - 1), -sH, H, atm(ws.DP16[1], xH, T - 1)},
                             {ws.pT[1], at(rs.P[0], sH, T - 1), atm(ws.DP[0], sH, T - 1), -sH, H, atm(ws.DP16[0], xH, T - 1)}};
      const SeqLink lq[2] = {{ws.qT[2], at(rs.Q[1], sH, T - 1), atm(ws.DQ[1], sH, T - 1), -sH, H, atm(ws.DQ16[1], xH, T - 1)},
                             {ws.qT[1], at(rs.Q[0], sH, T - 1), atm(ws.DQ[0], sH, T - 1), -sH, H, atm(ws.DQ16[0], xH, T - 1)}};
      add_linseq(bld, ctH, 0, half, H, false, true, 0, T, at(ws.DP16[2], xH, T - 1), -xH, 2, lp, -sH, -xH, ctH, slope, H);
      add_linseq(bld, ctH, half, half, H, false, true, 0, T, at(ws.DQ16[2], xH, T - 1), -xH, 2, lq, -sH, -xH, ctH, slope, H);
    } else {
      for (int k = 2; k >= 1; --k) {
        lin(ws.DP16[k], xH, ws.pT[k], H, rs.P[k - 1], ws.DP[k - 1], ws.DP16[k - 1], 0);
        lin(ws.DQ16[k], xH, ws.qT[k], H, rs.Q[k - 1], ws.DQ[k - 1], ws.DQ16[k - 1], half);
      }
    }
    if (d_z0) {  // s = T': gradient wrt the initial latent through both first layers of step 0 (two links: every word written once)
      const int r_z = range_for(ctZ * rt, half);
      Desc& d1 = bld.add(K_LIN, ctZ, 0, r_z, H, DF_RM_SC1, T, T + 1);
      bld.ptr(d1, 0, at(ws.DP16[0], xH, T), -xH); bld.ptr(d1, 1, ws.pzT); bld.ptr(d1, 5, ws.DZ0);
      d1.ld[3] = Z;
      Desc& d2 = bld.add(K_LIN, ctZ, half, r_z, H, DF_ADD_POLLED, T, T + 1);
      bld.ptr(d2, 0, at(ws.DQ16[0], xH, T), -xH); bld.ptr(d2, 1, ws.qzT); bld.ptr(d2, 3, ws.DZ0); bld.ptr(d2, 5, d_z0);
      d2.ld[1] = Z; d2.ld[3] = Z;
    }
    BLVM_REQUIRE(!bld.overflow, "srnn_bwd: persistent program overflow");
    rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
    if (rc) return rc;
    BLVM_HIP(pchain_fill_sentinel(ws.DZ0, (size_t)(reinterpret_cast<char*>(ws.x16_end) - reinterpret_cast<char*>(ws.DZ0)), s));
    TRY(pchain_launch(bld.p, s));
  }
  for (int t = Tp - 1; t >= 0 && !persistent; --t) {
    const size_t oH = (size_t)t * B * H, oZ = (size_t)t * B * Z, o2Z = (size_t)t * B * 2 * Z;
    // B1: dz_t = decoder gradient + the two first layers of step t+1, then rsample / residual / KL / softplus heads
    DzArgs dz;
    dz.has_gemm = t < Tp - 1;
    dz.D = ws.DP[0] + (dz.has_gemm ? oH + (size_t)B * H : 0); dz.WT = ws.pzT;
    dz.D2 = ws.DQ[0] + (dz.has_gemm ? oH + (size_t)B * H : 0); dz.WT2 = ws.qzT;
    dz.dz_add = d_z + oZ; dz.ld_add = Z;
    dz.mu_q = mu_q + oZ; dz.sd_q = sd_q + oZ; dz.mu_p = mu_p + oZ; dz.sd_p = sd_p + oZ; dz.eps = eps + oZ;
    dz.raw_q = rs.RAWQ + oZ; dz.raw_p = rs.RAWP + oZ;
    dz.x_sl = x_sl; dz.c_raw = c_raw; dz.c_fn = c_fn;
    dz.dqh = ws.DQH + o2Z; dz.dph = ws.DPH + o2Z;
    dz.B = B; dz.H = H; dz.Z = Z; dz.residual = residual_posterior; dz.t = t; dz.stride = stride;
    dz.fn_floor = fn_floor; dz.beta = beta; dz.sd_eps = sd_eps; dz.muq_raw = nullptr;
    launch_dz(dz, pick_nw(H, 2), dim3(Z / 16, rt), s);
    // B2: heads -> third layers;  B3, B4: down to the first layers (LeakyReLU derivatives fused)
    LinLaunch l;
    l.B = B; l.slope = slope; l.nseg = 2;
    l.seg[0] = seg(ws.DPH + o2Z, 2 * Z, ws.phT, 2 * Z, nullptr, nullptr, 0, rs.P[2] + oH, H, ws.DP[2] + oH, H, H, 2 * Z, 0);
    l.seg[1] = seg(ws.DQH + o2Z, 2 * Z, ws.qhT, 2 * Z, nullptr, nullptr, 0, rs.Q[2] + oH, H, ws.DQ[2] + oH, H, H, 2 * Z, 0);
    launch_lin(l, s);
    for (int k = 2; k >= 1; --k) {
      l.seg[0] = seg(ws.DP[k] + oH, H, ws.pT[k], H, nullptr, nullptr, 0, rs.P[k - 1] + oH, H, ws.DP[k - 1] + oH, H, H, H, 0);
      l.seg[1] = seg(ws.DQ[k] + oH, H, ws.qT[k], H, nullptr, nullptr, 0, rs.Q[k - 1] + oH, H, ws.DQ[k - 1] + oH, H, H, H, 0);
      launch_lin(l, s);
    }
  }
  BLVM_CHECK_LAUNCH("srnn_latent_bwd");
  if (d_z0 && !persistent) {  // gradient wrt the initial latent: both first layers of step 0
    LinLaunch l;
    l.B = B; l.slope = 0.f; l.nseg = 1;
    l.seg[0] = seg(ws.DP[0], H, ws.pzT, H, nullptr, nullptr, 0, nullptr, 0, d_z0, Z, Z, H, 0);
    launch_lin(l, s);
    l.seg[0] = seg(ws.DQ[0], H, ws.qzT, H, nullptr, d_z0, Z, nullptr, 0, d_z0, Z, Z, H, 0);
    launch_lin(l, s);
  }
  // batched, state-independent part
  if (d_d) TRY(gemm_f32(0, 1, (int)n, R, H, ws.DP[0], H, w->prior_w[0], ldw0, d_d, R, nullptr, 0, 0.f, nullptr, 0, 0, 1, s));
  if (d_a) TRY(gemm_f32(0, 1, (int)n, R, H, ws.DQ[0], H, w->post_w[0], ldw0, d_a, R, nullptr, 0, 0.f, nullptr, 0, 0, 1, s));
  WgradGroup grp;  // every weight gradient of the sequence: one grouped launch
  grp.add(ws.DP[0], H, H, d, R, R, gr->prior_w[0], ldw0);
  grp.add(ws.DP[0], H, H, zs, Z, Z, gr->prior_w[0] ? gr->prior_w[0] + R : nullptr, ldw0, gr->prior_b[0]);
  grp.add(ws.DQ[0], H, H, a, R, R, gr->post_w[0], ldw0);
  grp.add(ws.DQ[0], H, H, zs, Z, Z, gr->post_w[0] ? gr->post_w[0] + R : nullptr, ldw0, gr->post_b[0]);
  for (int k = 1; k < 3; ++k) {
    grp.add(ws.DP[k], H, H, rs.P[k - 1], H, H, gr->prior_w[k], H, gr->prior_b[k]);
    grp.add(ws.DQ[k], H, H, rs.Q[k - 1], H, H, gr->post_w[k], H, gr->post_b[k]);
  }
  grp.add(ws.DPH, 2 * Z, 2 * Z, rs.P[2], H, H, gr->prior_hw, H, gr->prior_hb);
  grp.add(ws.DQH, 2 * Z, 2 * Z, rs.Q[2], H, H, gr->post_hw, H, gr->post_hb);
  TRY(grp.run(n, s));
#undef TRY
  return BLVM_OK;
}
