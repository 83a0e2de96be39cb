// srnn_decode.hip — K3c: ancestral sampling from SRNNAudio, every step of every utterance in ONE persistent launch.
//
// Replaces the loop of `SRNN.generate` (blvm/models/srnn.py:304-403): enc = encoder(x_t) -> d_t = GRU(enc, d_{t-1}) (srnn.py:113) ->
// prior(cat[d_t, z_{t-1}]) -> z_t = mu + sd eps (srnn.py:92-111) -> decoder(cat[z_t, d_t]) -> DMoL head per sample -> draw -> x_{t+1}.
// A step is a program of 13 links for the persistent-chain engine (pchain.h / pchain.hip), every link's 16x16 tiles dealt over the
// whole chip; the 13.6 MB of weights stay in the L2s, activations travel as sentinel-polled T16 copies in per-step slabs.  The two
// concatenated inputs are ONE T16 slab each, written in parts by the links that produce the parts:
//   CP[s] = cat[d_s | z_{s-1}] (GRU link of step s, head link of step s-1; also the hidden-projection input of step s+1),
//   DC[s] = cat[z_s | d_s]     (head link, GRU link).
#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {
constexpr int SD_F = 30, SD_K = 10;  // DMoL head: 3 * num_mix parameters per sample

struct SdPack { size_t enc[3], wih, whh, prior[3], prior_h, dec[3], total; };
SdPack sd_pack_layout(int S, int H, int Z, int R) {
  SdPack p;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
  p.enc[0] = take((size_t)H * S); p.enc[1] = take((size_t)H * H); p.enc[2] = take((size_t)H * H);
  p.wih = take((size_t)3 * R * H); p.whh = take((size_t)3 * R * R);
  p.prior[0] = take((size_t)H * (R + Z)); p.prior[1] = take((size_t)H * H); p.prior[2] = take((size_t)H * H);
  p.prior_h = take((size_t)2 * Z * H);
  p.dec[0] = take((size_t)H * (Z + R)); p.dec[1] = take((size_t)H * H); p.dec[2] = take((size_t)S * SD_F * H);
  p.total = o;
  return p;
}
struct SdBufs { size_t X16, E16[2], ENC16, CP16, DS, GHb, P16[3], DC16, D16[2], DEC, ZS, dummyZ, dummyR, end; };
SdBufs sd_layout(size_t base, int T, int B, int S, int H, int Z, int R) {
  SdBufs b;
  size_t o = base;
  auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
  const size_t rows = (size_t)((B + 15) / 16) * 16, m = (size_t)T * rows;
  b.X16 = take((m + rows) * S);
  b.E16[0] = take(m * H); b.E16[1] = take(m * H); b.ENC16 = take(m * H);
  b.CP16 = take((m + 2 * rows) * (R + Z));
  b.DS = take((size_t)(T + 1) * B * R);
  b.GHb = take((size_t)T * B * 3 * R);
  for (int i = 0; i < 3; ++i) b.P16[i] = take(m * H);
  b.DC16 = take(m * (Z + R));
  b.D16[0] = take(m * H); b.D16[1] = take(m * H);
  b.DEC = take((size_t)T * B * S * SD_F);
  b.ZS = take((size_t)T * B * Z);  // z_t row-major (an output)
  b.dummyZ = take((size_t)B * Z);
  b.dummyR = take((size_t)B * R);
  b.end = o;
  return b;
}
}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_srnn_generate_scratch_floats(int T, int B, int S, int H, int Z, int R) {
  if (T <= 0 || B <= 0 || S <= 0 || H <= 0 || Z <= 0 || R <= 0) return 0;
  return sd_layout(sd_pack_layout(S, H, Z, R).total, T, B, S, H, Z, R).end;
}

extern "C" int blvm_srnn_generate(const BlvmSrnnDecodeWeights* w, const float* x0, const float* d0, const float* z0, const float* eps, const float* u,
                                  const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps, float slope, float log_eps,
                                  float* x_out, float* d_out, float* z_out, float* scratch, void* stream_) {
  using namespace pchain;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(w && w->chain && x0 && eps && x_out && scratch, "srnn_generate: null pointer");
  BLVM_REQUIRE(T >= 0 && B > 0 && B <= kPchainCarveMaxB, "srnn_generate: bad T=%d B=%d (at most %d utterances)", T, B, kPchainCarveMaxB);
  BLVM_REQUIRE(S % 16 == 0 && H % 16 == 0 && Z % 16 == 0 && R % 16 == 0 && S > 0 && H > 0 && Z > 0 && R > 0,
               "srnn_generate: S, H, Z, R must be positive multiples of 16 (got %d, %d, %d, %d)", S, H, Z, R);
  BLVM_REQUIRE(num_mix == SD_K, "srnn_generate: the DMoL head has %d components", SD_K);
  BLVM_REQUIRE((u == nullptr) == (v == nullptr), "srnn_generate: u and v are given together (both NULL: the mode)");
  BLVM_REQUIRE(aligned16(scratch), "srnn_generate: scratch must be 16-byte aligned");
  BLVM_REQUIRE(device_cus() >= 32, "srnn_generate: needs a device with at least 32 CUs");
  if (T == 0) return BLVM_OK;
  const BlvmSrnnWeights* c = w->chain;
  const SdPack p = sd_pack_layout(S, H, Z, R);
  const SdBufs b = sd_layout(p.total, T, B, S, H, Z, R);
  T16PackScope pack_scope(pchain_bf16(B), s);
  int rc;
#define PACK(dst, src, ld, rows, k)                               \
  do {                                                            \
    rc = t16_pack_rows(src, ld, rows, k, scratch + (dst), s);     \
    if (rc) return rc;                                            \
  } while (0)
  PACK(p.enc[0], w->enc_w[0], S, H, S); PACK(p.enc[1], w->enc_w[1], H, H, H); PACK(p.enc[2], w->enc_w[2], H, H, H);
  PACK(p.wih, w->gru_wih, H, 3 * R, H); PACK(p.whh, w->gru_whh, R, 3 * R, R);
  PACK(p.prior[0], c->prior_w[0], R + Z, H, R + Z); PACK(p.prior[1], c->prior_w[1], H, H, H); PACK(p.prior[2], c->prior_w[2], H, H, H);
  PACK(p.prior_h, c->prior_hw, H, 2 * Z, H);
  PACK(p.dec[0], w->dec_w[0], Z + R, H, Z + R); PACK(p.dec[1], w->dec_w[1], H, H, H); PACK(p.dec[2], w->dec_w[2], H, S * SD_F, H);
#undef PACK
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;
  const int rt = (B + 15) / 16, ctS = S / 16, ctH = H / 16, ctZ = Z / 16, ctR = R / 16, cus = device_cus() & ~7;
  const int nCP = (R + Z) / 16, nDC = (Z + R) / 16;
  const long rows = (long)rt * 16, xS = rows * S, xH = rows * H, xCP = rows * (R + Z), xDC = rows * (Z + R);
  const long sR = (long)B * R, s3R = 3 * sR, sZ = (long)B * Z, sF = (long)B * S * SD_F;
  float* const sc = scratch;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
  const int r_side = range_for(3 * ctR * rt, std::min(cus / 4, 64));  // the hidden projection of the NEXT step: off the critical path
  const int r_main = cus - r_side;
  Builder bld;
  bld.p.bf16 = pchain_bf16(B);
  bld.p.S = T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 4;
  bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = r_main;
  // out = leaky(A W^T + bias): A a polled T16 slab of `a_n16` blocks per row tile, outputs: T16 slab(s) and / or row-major (polled words)
  auto lin = [&](size_t A16, long a_step, int a_n16, size_t W, int K, const float* bias, int ct, int flags, float* orm, long rm_step, int ldo, size_t o16,
                 long o16_step, int n16, int wg0, int nwg) {
    Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags, 0, T);
    bld.ptr(d, 0, sc + A16, a_step); bld.ptr(d, 1, sc + W); bld.ptr(d, 2, bias); bld.ptr(d, 5, orm, rm_step);
    bld.ptr(d, 6, o16 ? sc + o16 : nullptr, o16_step);
    d.ld[0] = a_n16 * 16; d.ld[3] = ldo; d.n16[0] = n16; d.f[0] = slope;
  };
  const int rH = range_for(ctH * rt, r_main);
  // CP slab index: slab 0 = [d_0 | -], slab s + 1 = cat[d_s | z_{s-1}] of step s
  lin(b.X16, xS, ctS, p.enc[0], S, w->enc_b[0], ctH, DF_RELU, nullptr, 0, 0, b.E16[0], xH, ctH, 0, rH);
  const bool seq = linseq_enabled();  // runs of links of one shape as one descriptor (K_LINSEQ)
  if (seq) {
    const SeqLink le[2] = {{sc + p.enc[1], w->enc_b[1], nullptr, 0, 0, sc + b.E16[1]}, {sc + p.enc[2], w->enc_b[2], nullptr, 0, 0, sc + b.ENC16}};
    add_linseq(bld, ctH, 0, rH, H, true, false, 0, T, sc + b.E16[0], xH, 2, le, 0, xH, ctH, slope, 0);
  } else {
    lin(b.E16[0], xH, ctH, p.enc[1], H, w->enc_b[1], ctH, DF_RELU, nullptr, 0, 0, b.E16[1], xH, ctH, 0, rH);
    lin(b.E16[1], xH, ctH, p.enc[2], H, w->enc_b[2], ctH, DF_RELU, nullptr, 0, 0, b.ENC16, xH, ctH, 0, rH);
  }
  // gh_s = d_{s-1} Whh^T + b_hh: reads the d-part of slab s, first needed by the GRU link's epilogue
  lin(b.CP16, xCP, nCP, p.whh, R, w->gru_bhh, 3 * ctR, DF_RM_SC1 | DF_GENTLE | ((pchain_tune() & 16) ? DF_CANARY : 0), sc + b.GHb, s3R, 3 * R, 0, 0, 0, r_main,
      r_side);
  {  // d_s = GRU(enc_s, d_{s-1})
    Desc& d = bld.add(K_GRU, ctR, 0, range_for(ctR * rt, r_main), H, 0, 0, T);
    bld.ptr(d, 0, sc + b.ENC16, xH); bld.ptr(d, 1, sc + p.wih); bld.ptr(d, 2, nullptr); bld.ptr(d, 3, sc + b.GHb, s3R); bld.ptr(d, 4, sc + b.DS, sR);
    bld.ptr(d, 5, sc + b.DS + sR, sR); bld.ptr(d, 6, sc + b.CP16 + xCP, xCP); bld.ptr(d, 7, sc + b.dummyR); bld.ptr(d, 8, sc + b.dummyR);
    bld.ptr(d, 9, sc + b.dummyR); bld.ptr(d, 10, w->gru_bih); bld.ptr(d, 11, sc + b.DC16 + (size_t)ctZ * 256, xDC);
    d.ld[0] = R; d.ld[3] = R; d.n16[0] = nCP; d.n16[1] = nDC; d.i[0] = R;
  }
  // prior(cat[d_s, z_{s-1}])
  lin(b.CP16 + xCP, xCP, nCP, p.prior[0], R + Z, c->prior_b[0], ctH, DF_RELU, nullptr, 0, 0, b.P16[0], xH, ctH, 0, rH);
  if (seq) {
    const SeqLink lp[2] = {{sc + p.prior[1], c->prior_b[1], nullptr, 0, 0, sc + b.P16[1]}, {sc + p.prior[2], c->prior_b[2], nullptr, 0, 0, sc + b.P16[2]}};
    add_linseq(bld, ctH, 0, rH, H, true, false, 0, T, sc + b.P16[0], xH, 2, lp, 0, xH, ctH, slope, 0);
  } else {
    lin(b.P16[0], xH, ctH, p.prior[1], H, c->prior_b[1], ctH, DF_RELU, nullptr, 0, 0, b.P16[1], xH, ctH, 0, rH);
    lin(b.P16[1], xH, ctH, p.prior[2], H, c->prior_b[2], ctH, DF_RELU, nullptr, 0, 0, b.P16[2], xH, ctH, 0, rH);
  }
  {  // z_s ~ prior: into the decoder input and into the NEXT step's prior input
    Desc& d = bld.add(K_HEAD, ctZ, 0, range_for(ctZ * rt, r_main), H, 0, 0, T);
    bld.ptr(d, 0, sc + b.P16[2], xH); bld.ptr(d, 1, sc + b.P16[2], xH); bld.ptr(d, 2, sc + p.prior_h); bld.ptr(d, 3, c->prior_hb);
    bld.ptr(d, 4, sc + p.prior_h); bld.ptr(d, 5, c->prior_hb); bld.ptr(d, 6, eps, sZ);
    for (int k = 7; k <= 12; ++k) bld.ptr(d, k, sc + b.dummyZ);
    bld.ptr(d, 13, nullptr); bld.ptr(d, 14, sc + b.ZS, sZ); bld.ptr(d, 15, sc + b.DC16, xDC);
    bld.ptr(d, 16, sc + b.CP16 + 2 * xCP + (size_t)ctR * 256, xCP);
    d.ld[3] = Z; d.n16[0] = nDC; d.n16[1] = nCP; d.i[0] = Z; d.i[1] = 3; d.f[0] = beta; d.f[1] = 1.f / beta; d.f[2] = sd_eps;
  }
  // decoder(cat[z_s, d_s]); the last layer (S * F columns) on every workgroup
  lin(b.DC16, xDC, nDC, p.dec[0], Z + R, w->dec_b[0], ctH, DF_RELU, nullptr, 0, 0, b.D16[0], xH, ctH, 0, rH);
  lin(b.D16[0], xH, ctH, p.dec[1], H, w->dec_b[1], ctH, DF_RELU, nullptr, 0, 0, b.D16[1], xH, ctH, 0, rH);
  lin(b.D16[1], xH, ctH, p.dec[2], H, w->dec_b[2], S * SD_F / 16, DF_RELU | DF_RM_SC1, sc + b.DEC, sF, S * SD_F, 0, 0, 0, 0, range_for(S * SD_F / 16 * rt, cus));
  {  // per sample: head Linear -> DMoL draw -> x_{s+1}
    Desc& d = bld.add(K_DMOLS, S / 4, 0, range_for(S / 4 * rt, r_main), 16, 0, 0, T);
    bld.ptr(d, 0, sc + b.DEC, sF); bld.ptr(d, 1, w->lik_w); bld.ptr(d, 2, w->lik_b); bld.ptr(d, 3, u, (long)B * S * SD_K); bld.ptr(d, 4, v, (long)B * S);
    bld.ptr(d, 5, x_out, S); bld.ptr(d, 6, sc + b.X16 + xS, xS);
    d.ld[0] = S * SD_F; d.ld[3] = T * S; d.n16[0] = ctS; d.i[0] = S; d.i[1] = SD_F; d.i[2] = SD_K; d.f[0] = log_eps;
  }
  BLVM_REQUIRE(!bld.overflow, "srnn_generate: persistent program overflow");
  rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
  if (rc) return rc;
  // sentinel-fill everything the launch polls, then the initial frame stack and states
  BLVM_HIP(pchain_fill_sentinel(sc + b.X16, sizeof(float) * (b.ZS - b.X16), s));
  rc = pchain_rows_to_t16(x0, S, B, S, sc + b.X16, s); if (rc) return rc;
  rc = pchain_rows_to_t16(d0, R, B, R, sc + b.CP16, s, nCP); if (rc) return rc;
  rc = pchain_rows_to_t16(z0, Z, B, Z, sc + b.CP16 + xCP + (size_t)ctR * 256, s, nCP); if (rc) return rc;
  if (d0) BLVM_HIP(hipMemcpyAsync(sc + b.DS, d0, sizeof(float) * (size_t)B * R, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(sc + b.DS, 0, sizeof(float) * (size_t)B * R, s));
  rc = pchain_launch(bld.p, s);
  if (rc) return rc;
  if (d_out) BLVM_HIP(hipMemcpyAsync(d_out, sc + b.DS + (size_t)T * sR, sizeof(float) * (size_t)B * R, hipMemcpyDeviceToDevice, s));
  if (z_out) BLVM_HIP(hipMemcpyAsync(z_out, sc + b.ZS, sizeof(float) * (size_t)T * B * Z, hipMemcpyDeviceToDevice, s));
  return BLVM_OK;
}
