// kl.hip — K8: fused analytic Gaussian KL + free-nats floor + stride mask + per-utterance fp64 sums.
//
// Replaces kl_divergence_gaussian (blvm/utils/variational.py:67-70), discount_free_nats (:86-122, shared over the
// last dim) and the masked sums of VRNN/SRNN.compute_elbo (blvm/models/vrnn.py:271-276, srnn.py:150-156).
// HBM-bound: 4 x 4 B read per latent element forward (64 B per audio frame at z=256, s=64 — SURVEY.md §8d).
//
// Mapping (forward): a workgroup owns ONE utterance and a chunk of its latent steps — the four waves take steps in turn, a lane
// 4 consecutive latent dimensions (16-byte loads: a 1 KB row per wave instruction at Z = 256) — and accumulates both sums in
// float64 registers across its rows; wave shuffle + LDS reduction, then ONE fp64 atomic per workgroup and output.  (Round 1 had one
// wave per row with an atomic per row and output: 32 000 fp64 atomics on 128 addresses at [64,16000] — 59 us for 65.5 MB, 14 % of
// the HBM rate; the atomics, not the bytes, were the time.)  Backward: one wave per row, elementwise.
#include "common.h"

namespace blvm {
namespace {

struct KlArgs {
  const float *mu_q, *sd_q, *mu_p, *sd_p;
  const int32_t* x_sl;
  const float *c_raw, *c_fn;
  double *kld, *kld_fn;
  float *d_mu_q, *d_sd_q, *d_mu_p, *d_sd_p;
  int layout, B, Tp, Z, stride;
  float fn_floor;
};

__device__ __forceinline__ void row_coord(const KlArgs& a, int row, int& b, int& t) {
  if (a.layout == 0) { b = row / a.Tp; t = row - b * a.Tp; }
  else { t = row / a.B; b = row - t * a.B; }
}

__device__ __forceinline__ float kl_elem(float mq, float sq, float mp, float sp) {
  const float d = mq - mp;
  return log_(sp) - log_(sq) + (sq * sq + d * d) / (2.f * sp * sp) - 0.5f;
}

// grid (B, ceil(Tp / chunk)); VEC: Z % 4 == 0 and 16-byte aligned rows
template <bool VEC>
__global__ __launch_bounds__(256) void kl_fwd_kernel(KlArgs a, int chunk) {
  __shared__ double part[2][4];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = blockIdx.y * chunk, t1 = min(a.Tp, t0 + chunk);
  const long long xs = a.x_sl[b];
  const bool use_fn = a.fn_floor > 0.f;
  double s_raw = 0.0, s_fn = 0.0;
  auto row_base = [&](int t) { return ((a.layout == 0) ? (size_t)b * a.Tp + t : (size_t)t * a.B + b) * a.Z; };
  auto live = [&](int t) { return t < t1 && (long long)t * a.stride < xs; };  // masked steps contribute nothing
  if (VEC) {
    // four rows (one per wave stride) in flight per wave: 16 independent 16-byte loads per lane before anything waits — with one
    // row at a time a wave paid one memory round trip per 4 KB and the kernel ran at 15 % of the HBM rate
    constexpr int R = 4;
    for (int tb = t0 + wave; live(tb); tb += 4 * R) {
      for (int c = lane * 4; c < a.Z; c += 256) {
        float4 mq[R], sq[R], mp[R], sp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int t = tb + 4 * r;
          const size_t base = row_base(live(t) ? t : tb) + c;
          mq[r] = *reinterpret_cast<const float4*>(a.mu_q + base); sq[r] = *reinterpret_cast<const float4*>(a.sd_q + base);
          mp[r] = *reinterpret_cast<const float4*>(a.mu_p + base); sp[r] = *reinterpret_cast<const float4*>(a.sd_p + base);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!live(tb + 4 * r)) continue;
          const float k0 = kl_elem(mq[r].x, sq[r].x, mp[r].x, sp[r].x), k1 = kl_elem(mq[r].y, sq[r].y, mp[r].y, sp[r].y);
          const float k2 = kl_elem(mq[r].z, sq[r].z, mp[r].z, sp[r].z), k3 = kl_elem(mq[r].w, sq[r].w, mp[r].w, sp[r].w);
          s_raw += (double)k0; s_raw += (double)k1; s_raw += (double)k2; s_raw += (double)k3;
          if (use_fn) { s_fn += (double)fmaxf(k0, a.fn_floor); s_fn += (double)fmaxf(k1, a.fn_floor); s_fn += (double)fmaxf(k2, a.fn_floor); s_fn += (double)fmaxf(k3, a.fn_floor); }
        }
      }
    }
  } else {
    for (int t = t0 + wave; live(t); t += 4) {
      const size_t base = row_base(t);
      for (int c = lane; c < a.Z; c += 64) {
        const float k = kl_elem(a.mu_q[base + c], a.sd_q[base + c], a.mu_p[base + c], a.sd_p[base + c]);
        s_raw += (double)k;
        if (use_fn) s_fn += (double)fmaxf(k, a.fn_floor);
      }
    }
  }
  if (!use_fn) s_fn = s_raw;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_raw += __shfl_down(s_raw, off, 64);
    s_fn += __shfl_down(s_fn, off, 64);
  }
  if (lane == 0) { part[0][wave] = s_raw; part[1][wave] = s_fn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double r = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]), f = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    if (r != 0.0 || f != 0.0) {
      atomicAdd(a.kld + b, r);
      atomicAdd(a.kld_fn + b, f);
    }
  }
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(KlArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= a.B * a.Tp) return;
  int b, t;
  row_coord(a, row, b, t);
  const bool live = (long long)t * a.stride < a.x_sl[b];
  const float cr = (live && a.c_raw) ? a.c_raw[b] : 0.f;
  const float cf = (live && a.c_fn) ? a.c_fn[b] : 0.f;
  const size_t base = (size_t)row * a.Z;
  const bool use_fn = a.fn_floor > 0.f;
  for (int c = lane; c < a.Z; c += 64) {
    const float mq = a.mu_q[base + c], sq = a.sd_q[base + c], mp = a.mu_p[base + c], sp = a.sd_p[base + c];
    float coef = cr;
    if (cf != 0.f) {
      // d max(kl, floor)/d kl: 1 above the floor, 0 below (ties have measure zero)
      const float k = kl_elem(mq, sq, mp, sp);
      coef += (!use_fn || k > a.fn_floor) ? cf : 0.f;
    }
    const float d = mq - mp, ip2 = 1.f / (sp * sp);
    a.d_mu_q[base + c] = coef * d * ip2;
    a.d_mu_p[base + c] = -coef * d * ip2;
    a.d_sd_q[base + c] = coef * (sq * ip2 - 1.f / sq);
    a.d_sd_p[base + c] = coef * (1.f / sp - (sq * sq + d * d) * ip2 / sp);
  }
}

int check(const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, int layout,
          const int32_t* x_sl, int B, int Tp, int Z, int stride) {
  BLVM_REQUIRE(mu_q && sd_q && mu_p && sd_p && x_sl, "kl: null pointer");
  BLVM_REQUIRE(B > 0 && Tp > 0 && Z > 0 && stride > 0, "kl: bad shape B=%d Tp=%d Z=%d stride=%d", B, Tp, Z, stride);
  BLVM_REQUIRE(layout == 0 || layout == 1, "kl: layout must be 0 or 1");
  BLVM_REQUIRE((long long)B * Tp < (1ll << 31), "kl: too many rows");
  return BLVM_OK;
}

// ---- Gaussian latent head: softplus heads, posterior combination, reparameterised sample (elementwise) ---------------
struct LatentArgs {
  const float *mu_p, *sp_raw, *mq, *sq_raw, *eps;
  float *sd_p, *mu_q, *sd_q, *z;                          // forward outputs
  const float *g_sd_p, *g_mu_q, *g_sd_q, *g_z;            // backward: upstream gradients (each may be null)
  float *d_mu_p, *d_sp_raw, *d_mq, *d_sq_raw;             // backward outputs
  size_t n;
  float beta_p, beta_q, sd_eps;
  int mode;  // 0 plain, 1 residual (mu_q += mu_p), 2 precision-weighted
};

template <bool BWD>
__global__ __launch_bounds__(256) void latent_head_kernel(LatentArgs a) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
    const float mu_p = a.mu_p[i], mq = a.mq[i], rp = a.sp_raw[i], rq = a.sq_raw[i], e = a.eps[i];
    const float sp = softplus_beta(rp, a.beta_p, 1.f / a.beta_p) + a.sd_eps;
    const float sq0 = softplus_beta(rq, a.beta_q, 1.f / a.beta_q) + a.sd_eps;
    float mu_q, sd_q, pr_p = 0.f, pr_q = 0.f, var = 0.f;
    if (a.mode == 2) {
      pr_p = 1.f / (sp * sp);
      pr_q = 1.f / (sq0 * sq0);
      var = 1.f / (pr_p + pr_q);
      mu_q = var * (mu_p * pr_p + mq * pr_q);
      sd_q = sqrtf(var);
    } else {
      mu_q = a.mode == 1 ? mq + mu_p : mq;
      sd_q = sq0;
    }
    if (!BWD) {
      a.sd_p[i] = sp;
      a.mu_q[i] = mu_q;
      a.sd_q[i] = sd_q;
      a.z[i] = fmaf(sd_q, e, mu_q);
    } else {
      const float gz = a.g_z ? a.g_z[i] : 0.f;
      const float gm = (a.g_mu_q ? a.g_mu_q[i] : 0.f) + gz;
      const float gs = (a.g_sd_q ? a.g_sd_q[i] : 0.f) + gz * e;
      float d_sp = a.g_sd_p ? a.g_sd_p[i] : 0.f, d_sq0, d_mu_p, d_mq;
      if (a.mode == 2) {
        const float dvar = gm * (mu_p * pr_p + mq * pr_q) + gs / (2.f * sd_q);
        const float d_pr_p = gm * var * mu_p - dvar * var * var;
        const float d_pr_q = gm * var * mq - dvar * var * var;
        d_mu_p = gm * var * pr_p;
        d_mq = gm * var * pr_q;
        d_sp += d_pr_p * (-2.f * pr_p / sp);
        d_sq0 = d_pr_q * (-2.f * pr_q / sq0);
      } else {
        d_mu_p = a.mode == 1 ? gm : 0.f;
        d_mq = gm;
        d_sq0 = gs;
      }
      a.d_mu_p[i] = d_mu_p;
      a.d_mq[i] = d_mq;
      a.d_sp_raw[i] = d_sp * sigmoidf_(a.beta_p * rp);   // d softplus_beta(x)/dx = sigmoid(beta x)
      a.d_sq_raw[i] = d_sq0 * sigmoidf_(a.beta_q * rq);
    }
  }
}

}  // namespace
}  // namespace blvm

extern "C" int blvm_gauss_latent_fwd(const float* mu_p, const float* sd_p_raw, const float* mu_q_raw, const float* sd_q_raw,
                                     const float* eps, size_t n, float beta_p, float beta_q, float sd_eps, int mode,
                                     float* sd_p, float* mu_q, float* sd_q, float* z, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(mu_p && sd_p_raw && mu_q_raw && sd_q_raw && eps && sd_p && mu_q && sd_q && z, "gauss_latent_fwd: null pointer");
  BLVM_REQUIRE(mode >= 0 && mode <= 2 && beta_p > 0.f && beta_q > 0.f, "gauss_latent_fwd: bad mode / beta");
  if (n == 0) return BLVM_OK;
  LatentArgs a{};
  a.mu_p = mu_p; a.sp_raw = sd_p_raw; a.mq = mu_q_raw; a.sq_raw = sd_q_raw; a.eps = eps;
  a.sd_p = sd_p; a.mu_q = mu_q; a.sd_q = sd_q; a.z = z;
  a.n = n; a.beta_p = beta_p; a.beta_q = beta_q; a.sd_eps = sd_eps; a.mode = mode;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((latent_head_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("gauss_latent_fwd");
  return BLVM_OK;
}

extern "C" int blvm_gauss_latent_bwd(const float* mu_p, const float* sd_p_raw, const float* mu_q_raw, const float* sd_q_raw,
                                     const float* eps, const float* g_sd_p, const float* g_mu_q, const float* g_sd_q,
                                     const float* g_z, size_t n, float beta_p, float beta_q, float sd_eps, int mode,
                                     float* d_mu_p, float* d_sd_p_raw, float* d_mu_q_raw, float* d_sd_q_raw, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(mu_p && sd_p_raw && mu_q_raw && sd_q_raw && eps && d_mu_p && d_sd_p_raw && d_mu_q_raw && d_sd_q_raw,
               "gauss_latent_bwd: null pointer");
  BLVM_REQUIRE(mode >= 0 && mode <= 2 && beta_p > 0.f && beta_q > 0.f, "gauss_latent_bwd: bad mode / beta");
  if (n == 0) return BLVM_OK;
  LatentArgs a{};
  a.mu_p = mu_p; a.sp_raw = sd_p_raw; a.mq = mu_q_raw; a.sq_raw = sd_q_raw; a.eps = eps;
  a.g_sd_p = g_sd_p; a.g_mu_q = g_mu_q; a.g_sd_q = g_sd_q; a.g_z = g_z;
  a.d_mu_p = d_mu_p; a.d_sp_raw = d_sd_p_raw; a.d_mq = d_mu_q_raw; a.d_sq_raw = d_sd_q_raw;
  a.n = n; a.beta_p = beta_p; a.beta_q = beta_q; a.sd_eps = sd_eps; a.mode = mode;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((latent_head_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("gauss_latent_bwd");
  return BLVM_OK;
}

extern "C" int blvm_kl_fwd(const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, int layout,
                           const int32_t* x_sl, int B, int Tp, int Z, int stride, float fn_floor, double* kld,
                           double* kld_fn, void* stream) {
  using namespace blvm;
  int rc = check(mu_q, sd_q, mu_p, sd_p, layout, x_sl, B, Tp, Z, stride);
  if (rc) return rc;
  BLVM_REQUIRE(kld && kld_fn, "kl_fwd: null output");
  KlArgs a{};
  a.mu_q = mu_q; a.sd_q = sd_q; a.mu_p = mu_p; a.sd_p = sd_p; a.x_sl = x_sl;
  a.kld = kld; a.kld_fn = kld_fn;
  a.layout = layout; a.B = B; a.Tp = Tp; a.Z = Z; a.stride = stride; a.fn_floor = fn_floor;
  // chunk of steps per workgroup: ~1 000 workgroups, at least one step per wave
  int chunk = (int)(((long long)B * Tp + 1023) / 1024);
  chunk = chunk < 4 ? 4 : (chunk > 256 ? 256 : chunk);
  const dim3 grid(B, (Tp + chunk - 1) / chunk);
  const bool vec = Z % 4 == 0 && aligned16(mu_q) && aligned16(sd_q) && aligned16(mu_p) && aligned16(sd_p);
  if (vec) hipLaunchKernelGGL((kl_fwd_kernel<true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, chunk);
  else hipLaunchKernelGGL((kl_fwd_kernel<false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, chunk);
  BLVM_CHECK_LAUNCH("kl_fwd");
  return BLVM_OK;
}

extern "C" int blvm_kl_bwd(const float* mu_q, const float* sd_q, const float* mu_p, const float* sd_p, int layout,
                           const int32_t* x_sl, const float* c_raw, const float* c_fn, int B, int Tp, int Z,
                           int stride, float fn_floor, float* d_mu_q, float* d_sd_q, float* d_mu_p, float* d_sd_p,
                           void* stream) {
  using namespace blvm;
  int rc = check(mu_q, sd_q, mu_p, sd_p, layout, x_sl, B, Tp, Z, stride);
  if (rc) return rc;
  BLVM_REQUIRE(d_mu_q && d_sd_q && d_mu_p && d_sd_p, "kl_bwd: null output");
  KlArgs a{};
  a.mu_q = mu_q; a.sd_q = sd_q; a.mu_p = mu_p; a.sd_p = sd_p; a.x_sl = x_sl;
  a.c_raw = c_raw; a.c_fn = c_fn;
  a.d_mu_q = d_mu_q; a.d_sd_q = d_sd_q; a.d_mu_p = d_mu_p; a.d_sd_p = d_sd_p;
  a.layout = layout; a.B = B; a.Tp = Tp; a.Z = Z; a.stride = stride; a.fn_floor = fn_floor;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3((B * Tp + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  BLVM_CHECK_LAUNCH("kl_bwd");
  return BLVM_OK;
}
