// vrnn_decode.hip — K1c: ancestral sampling from VRNNAudio, every step of every utterance in ONE launch.
//
// Replaces the Python loop of VRNN.generate (blvm/models/vrnn.py:371-434): per step  enc = encoder(x_t)  ->  cell in
// prior-sampling mode (VRNNCell.generate, vrnn.py:143-164: prior MLP -> (mu, softplus sd) -> z = mu + sd eps -> phi_z MLP ->
// GRU([enc, phi_z], h))  ->  decoder(cat[phi_z, h_new])  ->  DMoL head per sample  ->  sample  ->  x_{t+1}.
// That is 17 dependent matrix-vector products per step with M = the batch; launched one by one a step costs ~0.45 ms of host
// and dispatch time for 12.5 MB of weights.  Here a workgroup owns 16 utterances (the M of v_mfma_f32_16x16x4_f32), keeps
// every activation in LDS and walks the layers itself: the only traffic is the weight stream (T16 operand layout: one
// contiguous 1 KB read per fragment), the noise and the samples — the decode step is bound by how fast one CU ingests weights.
#include "common.h"
#include "pchain.h"

namespace blvm {
namespace {

constexpr int VD_ROWS = 16, VD_NW = 8, VD_CHUNK = 8;  // utterances per workgroup, waves, samples per decoder chunk
constexpr int VD_F = 30, VD_K = 10;                   // DMoL head: 3 * num_mix parameters per sample

struct VDArgs {
  // T16 operand copies (common.h) and the biases
  const float *enc_w[3], *enc_b[3];
  const float *prior_w[3], *prior_b[3], *prior_hw, *prior_hb;
  const float *phi_w[4], *phi_b[4];
  const float *wih, *whh, *bih, *bhh;
  const float *dec_w[3], *dec_b[3];
  const float *lik_w, *lik_b;  // [30,30] row-major, [30]
  const float *x0, *h0, *eps, *u, *v;
  float *x_out, *h_out;
  int T, B, S, H, Z, R;
  float sd_eps, beta, slope, log_eps;
};

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

__global__ __launch_bounds__(VD_NW * 64) void vrnn_decode_kernel(VDArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int S = a.S, H = a.H, Z = a.Z, R = a.R, B = a.B;
  const int ldS = S + 4, ldH = H + 4, ldZ = Z + 4, ldR = R + 4, ldD = VD_CHUNK * VD_F + 4;
  float* sX = smem;                      // [16][S]   current frame stack
  float* sEnc = sX + VD_ROWS * ldS;      // [16][H]   encoder output (GRU input)
  float* sH = sEnc + VD_ROWS * ldH;      // [16][R]   recurrent state
  float* sT0 = sH + VD_ROWS * ldR;       // [16][H]   ping
  float* sT1 = sT0 + VD_ROWS * ldH;      // [16][H]   pong
  float* sZ = sT1 + VD_ROWS * ldH;       // [16][Z]
  float* sPhi = sZ + VD_ROWS * ldZ;      // [16][H]
  float* sDec = sPhi + VD_ROWS * ldH;    // [16][8*30] one chunk of the decoder's last layer
  float* sLik = sDec + VD_ROWS * ldD;    // head Linear [30,30] zero-padded to [32,32] in the T16 operand layout (1024) + bias (32)
  float* sPar = sLik + 1024 + 32;        // [16*8][32] head outputs of a chunk

  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, cc = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler too: the item bookkeeping stays scalar
  const int b0 = blockIdx.x * VD_ROWS;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < VD_ROWS * S; i += VD_NW * 64) {
    const int r = i / S, c = i - r * S;
    sX[r * ldS + c] = b0 + r < B ? a.x0[(size_t)(b0 + r) * S + c] : 0.f;
  }
  for (int i = tid; i < VD_ROWS * R; i += VD_NW * 64) {
    const int r = i / R, c = i - r * R;
    sH[r * ldR + c] = (a.h0 != nullptr && b0 + r < B) ? a.h0[(size_t)(b0 + r) * R + c] : 0.f;
  }
  for (int i = tid; i < 1024 + 32; i += VD_NW * 64) {
    float val = 0.f;
    if (i < 1024) {  // block (o / 16, k / 16), fragment of lane (o % 16) + 16 ((k % 16) / 4), element k % 4
      const int e = i & 3, ln = (i >> 2) & 63, blk = i >> 8;
      const int o = (blk >> 1) * 16 + (ln & 15), k = (blk & 1) * 16 + 4 * (ln >> 4) + e;
      if (o < VD_F && k < VD_F) val = a.lik_w[o * VD_F + k];
    } else if (i - 1024 < VD_F) {
      val = a.lik_b[i - 1024];
    }
    sLik[i] = val;
  }
  for (int i = tid; i < VD_ROWS * 4; i += VD_NW * 64) sDec[(i >> 2) * ldD + VD_CHUNK * VD_F + (i & 3)] = 0.f;  // the k = 30, 31 padding reads
  __syncthreads();

  // ---- the weight stream.  A layer is, per wave, a flat list of ITEMS = (column tile, input segment, batch of <= 16 k-chunks):
  // the 16 KB of weight fragments of the NEXT item are requested while the current item runs on the matrix pipe (two register
  // sets that swap roles by position in the loop body), across tile and segment boundaries alike — with one wave per SIMD
  // nothing else hides the latency of a read that misses L2.
  constexpr int NB = 8;  // k-chunks per item
  struct Item { const float* W; const float* A; const float *bA, *bB; int bstride, n, acc; bool last; int tile; };
  auto fetch = [&](float4 (&w)[NB], float (&bb)[6], const Item& I) {
    // the tile's epilogue biases travel with the fragments (loaded for every item: an epilogue-time load would expose a full
    // memory latency per tile)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      bb[k] = I.bA[k * I.bstride];
      bb[3 + k] = I.bB[k * I.bstride];
    }
    // unconditional loads (chunks beyond I.n re-read the last valid one; their A fragments are zeroed instead): the compiler
    // can then count the loads in flight, and the wait before an item's MFMAs leaves the NEXT item's loads outstanding
#pragma unroll
    for (int j = 0; j < NB; ++j) w[j] = *reinterpret_cast<const float4*>(I.W + 256 * (j < I.n ? j : I.n - 1));
  };
  auto mac = [&](const float4 (&w)[NB], const Item& I, f32x4& acc) {
    float4 x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      x[j] = *reinterpret_cast<const float4*>(I.A + 16 * (j < I.n ? j : I.n - 1));
      if (j >= I.n) x[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NB; j += 2) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].x, w[j].x, acc, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j + 1].x, w[j + 1].x, c1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].y, w[j].y, acc, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j + 1].y, w[j + 1].y, c1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].z, w[j].z, acc, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j + 1].z, w[j + 1].z, c1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].w, w[j].w, acc, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j + 1].w, w[j + 1].w, c1, 0, 0, 0);
    }
    acc += c1;
  };
  f32x4 acc[6];
  auto mac_sel = [&](const float4 (&w)[NB], const Item& I, auto nacc) {
    constexpr int NACC = decltype(nacc)::value;  // wave-uniform selection keeps the accumulators in registers
    if constexpr (NACC == 1) {
      mac(w, I, acc[0]);
    } else if constexpr (NACC == 2) {
      if (I.acc == 0) mac(w, I, acc[0]);
      else mac(w, I, acc[1]);
    } else {
      switch (I.acc) {
        case 0: mac(w, I, acc[0]); break;
        case 1: mac(w, I, acc[1]); break;
        case 2: mac(w, I, acc[2]); break;
        case 3: mac(w, I, acc[3]); break;
        case 4: mac(w, I, acc[4]); break;
        default: mac(w, I, acc[5]); break;
      }
    }
  };
  auto run = [&](int nit, auto item, auto epi, auto nacc) {
    float4 w0[NB], w1[NB];
    float bb0[6], bb1[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) acc[g] = zero4;
    if (nit > 0) {
      // every fetch is unconditional (past the end: the last item again), so that between a fetch and the MFMAs that use the
      // OTHER register set there is no branch with memory traffic — otherwise the compiler's wait before the MFMAs is vmcnt(0)
      // and drains the prefetch it should overlap
      Item I0 = item(0), I1 = I0;
      fetch(w0, bb0, I0);
#pragma nounroll
      for (int it = 0; it < nit; it += 2) {
        I1 = item(it + 1 < nit ? it + 1 : nit - 1);
        fetch(w1, bb1, I1);
        mac_sel(w0, I0, nacc);
        const bool l0 = I0.last;
        const int t0 = I0.tile;
        float be[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) be[k] = bb0[k];
        I0 = item(it + 2 < nit ? it + 2 : nit - 1);
        fetch(w0, bb0, I0);
        if (l0) epi(t0, be);
        if (it + 1 < nit) {
          mac_sel(w1, I1, nacc);
          if (I1.last) epi(I1.tile, bb1);
        }
      }
    }
    __syncthreads();
  };
  const int a_off = (lane & 15), a_q = 4 * q;  // A fragment: row lane & 15, k = 4 (lane >> 4) .. + 3 of a chunk
  // items of a plain layer: tiles of this wave x segments (in0 | in1) x batches of 256 k
  // one step = a short PROGRAM of ops walked by a single loop, so that the layer pipeline below exists once in the code (as 13
  // inlined copies the kernel was 100+ KB, more than the instruction cache holds)
  struct Op { const float* W; const float* bias; int in0, ld0, K0, in1, ld1, K1, out, ldo, N, kind; float slope; int pad; };
  enum { OP_DENSE = 0, OP_HEAD = 1, OP_GRU = 2, OP_SAMPLE = 3 };
  Op* sProg = reinterpret_cast<Op*>(sPar + VD_ROWS * VD_CHUNK * 32);
  const int N_OPS = 6 + 1 + 4 + 1 + 2 + 2 * (S / VD_CHUNK);  // <= 32 (checked on the host)
  if (tid == 0) {
    int n = 0;
    auto D = [&](const float* W, const float* bias, float* in0, int ld0, int K0, float* in1, int ld1, int K1, float* out, int ldo, int N, float slope) {
      Op o;
      o.W = W; o.bias = bias; o.in0 = (int)(in0 - smem); o.ld0 = ld0; o.K0 = K0; o.in1 = in1 ? (int)(in1 - smem) : 0; o.ld1 = ld1; o.K1 = K1;
      o.out = (int)(out - smem); o.ldo = ldo; o.N = N; o.kind = OP_DENSE; o.slope = slope; o.pad = 0;
      sProg[n++] = o;
    };
    auto K = [&](int kind, int arg) { Op o{}; o.kind = kind; o.N = arg; sProg[n++] = o; };
    D(a.enc_w[0], a.enc_b[0], sX, ldS, S, nullptr, 0, 0, sT0, ldH, H, a.slope);
    D(a.enc_w[1], a.enc_b[1], sT0, ldH, H, nullptr, 0, 0, sT1, ldH, H, a.slope);
    D(a.enc_w[2], a.enc_b[2], sT1, ldH, H, nullptr, 0, 0, sEnc, ldH, H, a.slope);
    D(a.prior_w[0], a.prior_b[0], sH, ldR, R, nullptr, 0, 0, sT0, ldH, H, 0.f);
    D(a.prior_w[1], a.prior_b[1], sT0, ldH, H, nullptr, 0, 0, sT1, ldH, H, 0.f);
    D(a.prior_w[2], a.prior_b[2], sT1, ldH, H, nullptr, 0, 0, sT0, ldH, H, 0.f);
    K(OP_HEAD, 0);
    D(a.phi_w[0], a.phi_b[0], sZ, ldZ, Z, nullptr, 0, 0, sT1, ldH, H, 0.f);
    D(a.phi_w[1], a.phi_b[1], sT1, ldH, H, nullptr, 0, 0, sT0, ldH, H, 0.f);
    D(a.phi_w[2], a.phi_b[2], sT0, ldH, H, nullptr, 0, 0, sT1, ldH, H, 0.f);
    D(a.phi_w[3], a.phi_b[3], sT1, ldH, H, nullptr, 0, 0, sPhi, ldH, H, 0.f);
    K(OP_GRU, 0);
    D(a.dec_w[0], a.dec_b[0], sPhi, ldH, H, sH, ldR, R, sT0, ldH, H, a.slope);
    D(a.dec_w[1], a.dec_b[1], sT0, ldH, H, nullptr, 0, 0, sT1, ldH, H, a.slope);
    for (int ch = 0; ch < S / VD_CHUNK; ++ch) {
      const int c0 = ch * VD_CHUNK * VD_F;
      D(a.dec_w[2] + (size_t)c0 * H, a.dec_b[2] + c0, sT1, ldH, H, nullptr, 0, 0, sDec, ldD, VD_CHUNK * VD_F, a.slope);
      K(OP_SAMPLE, ch);
    }
  }
  __syncthreads();
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto unip = [&](const float* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = uni((int)(unsigned)u), hi = uni((int)(unsigned)(u >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
  };
  auto dense = [&](const Op& o) {
    const int N = uni(o.N), ld0 = uni(o.ld0), K0 = uni(o.K0), ld1 = uni(o.ld1), K1 = uni(o.K1), ldo = uni(o.ldo);
    const float* W = unip(o.W);
    const float* bias = unip(o.bias);
    const float* in0 = smem + uni(o.in0);
    const float* in1 = smem + uni(o.in1);
    float* out = smem + uni(o.out);
    const float slope = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, o.slope)));
    const int Kt = K0 + K1, nb0 = (K0 + 16 * NB - 1) / (16 * NB), nb1 = (K1 + 16 * NB - 1) / (16 * NB), ipt = nb0 + nb1;
    const int my_tiles = (N / 16 - wave + VD_NW - 1) / VD_NW;
    auto item = [&](int it) {
      Item I;
      const int ti = it / ipt, r = it - ti * ipt;
      I.tile = wave + ti * VD_NW;
      const bool second = r >= nb0;
      const int kb = second ? r - nb0 : r, Ks = second ? K1 : K0;
      const int k0 = kb * 16 * NB;
      I.n = (Ks - k0 < 16 * NB ? Ks - k0 : 16 * NB) / 16;
      I.W = W + (size_t)I.tile * 16 * Kt + (size_t)((second ? K0 : 0) + k0) * 16 + 4 * lane;
      I.A = (second ? in1 + a_off * ld1 : in0 + a_off * ld0) + a_q + k0;
      I.acc = 0;
      I.last = r == ipt - 1;
      I.bA = I.bB = bias + I.tile * 16 + cc;
      I.bstride = 0;
      return I;
    };
    run(my_tiles > 0 ? my_tiles * ipt : 0, item, [&](int tile, const float (&bb)[6]) {
      const float bv = bb[0];
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(4 * q + r) * ldo + tile * 16 + cc] = leaky(acc[0][r] + bv, slope);
      acc[0] = zero4;
    }, std::integral_constant<int, 1>{});
  };

#ifdef VD_PROF
  long long tk = __builtin_readcyclecounter(), ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define VD_TICK(k) { const long long n__ = __builtin_readcyclecounter(); ph[k] += n__ - tk; tk = n__; }
#else
#define VD_TICK(k)
#endif
  for (int t = 0; t < a.T; ++t) {
#pragma nounroll
    for (int op = 0; op < N_OPS; ++op) {
      const int kind = uni(sProg[op].kind);
      if (kind == OP_DENSE) {
        dense(sProg[op]);
      } else if (kind == OP_HEAD) {
        {  // head: mean tile (acc 0) and raw-scale tile (acc 1) of the same columns
          const int nbh = (H + 16 * NB - 1) / (16 * NB), ipt = 2 * nbh;
          const int my_tiles = (Z / 16 - wave + VD_NW - 1) / VD_NW;
          auto item = [&](int it) {
            Item I;
            const int ti = it / ipt, r = it - ti * ipt;
            I.tile = wave + ti * VD_NW;
            const int half = r / nbh, kb = r - half * nbh, k0 = kb * 16 * NB;
            I.n = (H - k0 < 16 * NB ? H - k0 : 16 * NB) / 16;
            I.W = a.prior_hw + (size_t)(half * Z + I.tile * 16) * H + (size_t)k0 * 16 + 4 * lane;
            I.A = sT0 + a_off * ldH + a_q + k0;
            I.acc = half;
            I.last = r == ipt - 1;
            I.bA = I.bB = a.prior_hb + I.tile * 16 + cc;
            I.bstride = Z;
            return I;
          };
          run(my_tiles > 0 ? my_tiles * ipt : 0, item, [&](int tile, const float (&bb)[6]) {
            const int col = tile * 16 + cc;
            const float bm = bb[0], br = bb[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = b0 + 4 * q + r < B ? a.eps[((size_t)t * B + b0 + 4 * q + r) * Z + col] : 0.f;
              const float sd = softplus_beta(acc[1][r] + br, a.beta, 1.f / a.beta) + a.sd_eps;
              sZ[(4 * q + r) * ldZ + col] = e * sd + (acc[0][r] + bm);
            }
            acc[0] = zero4; acc[1] = zero4;
          }, std::integral_constant<int, 2>{});
        }
      } else if (kind == OP_GRU) {
        {
          const int KI = 2 * H;  // X = H in VRNNAudio
          const int nbi = (H + 16 * NB - 1) / (16 * NB), nbh = (R + 16 * NB - 1) / (16 * NB), ipg = 2 * nbi + nbh, ipt = 3 * ipg;
          const int my_tiles = (R / 16 - wave + VD_NW - 1) / VD_NW;
          float hn[4][4];
          auto item = [&](int it) {
            Item I;
            const int ti = it / ipt, r = it - ti * ipt, g = r / ipg, p = r - g * ipg;
            I.tile = wave + ti * VD_NW;
            const size_t wrow = (size_t)(g * R + I.tile * 16);
            if (p < 2 * nbi) {  // input projection: enc then phi
              const int seg = p / nbi, kb = p - seg * nbi, k0 = kb * 16 * NB;
              I.n = (H - k0 < 16 * NB ? H - k0 : 16 * NB) / 16;
              I.W = a.wih + wrow * KI + (size_t)(seg * H + k0) * 16 + 4 * lane;
              I.A = (seg ? sPhi : sEnc) + a_off * ldH + a_q + k0;
              I.acc = g;
            } else {
              const int kb = p - 2 * nbi, k0 = kb * 16 * NB;
              I.n = (R - k0 < 16 * NB ? R - k0 : 16 * NB) / 16;
              I.W = a.whh + wrow * R + (size_t)k0 * 16 + 4 * lane;
              I.A = sH + a_off * ldR + a_q + k0;
              I.acc = 3 + g;
            }
            I.last = r == ipt - 1;
            I.bA = a.bih + I.tile * 16 + cc;
            I.bB = a.bhh + I.tile * 16 + cc;
            I.bstride = R;
            return I;
          };
          int nt = 0;
          // no barrier inside run() may separate the last read of sH from its update: run() ends with one
          run(my_tiles > 0 ? my_tiles * ipt : 0, item, [&](int tile, const float (&bb)[6]) {
            const int col = tile * 16 + cc;
            const float bir = bb[0], biu = bb[1], bin = bb[2];
            const float bhr = bb[3], bhu = bb[4], bhn = bb[5];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float rg = sigmoidf_(acc[0][r] + bir + acc[3][r] + bhr);
              const float ug = sigmoidf_(acc[1][r] + biu + acc[4][r] + bhu);
              const float ng = tanhf(acc[2][r] + bin + rg * (acc[5][r] + bhn));
              const float hv = (1.f - ug) * ng + ug * sH[(4 * q + r) * ldR + col];
              switch (nt) {  // static register indices
                case 0: hn[0][r] = hv; break;
                case 1: hn[1][r] = hv; break;
                case 2: hn[2][r] = hv; break;
                default: hn[3][r] = hv; break;
              }
            }
            ++nt;
#pragma unroll
            for (int g = 0; g < 6; ++g) acc[g] = zero4;
          }, std::integral_constant<int, 6>{});
          nt = 0;
          for (int tile = wave; tile < R / 16; tile += VD_NW, ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float hv;
              switch (nt) { case 0: hv = hn[0][r]; break; case 1: hv = hn[1][r]; break; case 2: hv = hn[2][r]; break; default: hv = hn[3][r]; break; }
              sH[(4 * q + r) * ldR + tile * 16 + cc] = hv;
            }
          __syncthreads();
        }
      } else {
        const int ch = uni(sProg[op].N);
        // the sampler's draws of this chunk travel while the head runs
        float uu[VD_K], vv = 0.5f;
        if (tid < VD_ROWS * VD_CHUNK) {
          const int r = tid / VD_CHUNK, sm = tid - r * VD_CHUNK;
          const size_t f = ((size_t)t * B + (b0 + r < B ? b0 + r : 0)) * S + ch * VD_CHUNK + sm;
          if (a.u != nullptr) {
#pragma unroll
            for (int m = 0; m < VD_K; ++m) uu[m] = a.u[f * VD_K + m];
            vv = a.v[f];
          }
        }
        // head Linear [30,30] per (utterance, sample) on the matrix pipe: K and N padded to 32 with zero weights; (sample, column
        // tile) pairs over the waves.  A fragments start at 30 sm + 16 j + 4 q floats: 8-byte aligned
        for (int w2 = wave; w2 < VD_CHUNK * 2; w2 += VD_NW) {
          const int sm = w2 >> 1, ct = w2 & 1;
          f32x4 c = zero4;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float* ap = sDec + (lane & 15) * ldD + sm * VD_F + 16 * j + 4 * q;
            const float2 x01 = *reinterpret_cast<const float2*>(ap), x23 = *reinterpret_cast<const float2*>(ap + 2);
            const float4 wf = *reinterpret_cast<const float4*>(sLik + ((ct * 2 + j) * 64 + lane) * 4);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x01.x, wf.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x01.y, wf.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x23.x, wf.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(x23.y, wf.w, c, 0, 0, 0);
          }
          const float bv = sLik[1024 + ct * 16 + cc];
#pragma unroll
          for (int r = 0; r < 4; ++r) sPar[((4 * q + r) * VD_CHUNK + sm) * 32 + ct * 16 + cc] = c[r] + bv;
        }
        __syncthreads();
        if (tid < VD_ROWS * VD_CHUNK) {  // Gumbel-max component pick + clamped logistic draw (as mix_sample_kernel, dmol.hip)
          const int r = tid / VD_CHUNK, sm = tid - r * VD_CHUNK, s_idx = ch * VD_CHUNK + sm;
          if (b0 + r < B) {
            const float* p = sPar + tid * 32;
            const size_t f = ((size_t)t * B + b0 + r) * S + s_idx;
            int best = 0;
            float bvv = -INFINITY;
            for (int m = 0; m < VD_K; ++m) {
              float sc = p[m];
              if (a.u != nullptr) sc -= logf(-logf(uu[m]));
              if (sc > bvv) { bvv = sc; best = m; }
            }
            float x = p[VD_K + best];
            if (a.v != nullptr) {
              x += expf(fmaxf(p[2 * VD_K + best], a.log_eps)) * (logf(vv) - logf(1.f - vv));
              x = fminf(fmaxf(x, -1.f), 1.f);
            }
            a.x_out[((size_t)(b0 + r) * a.T + t) * S + s_idx] = x;
            sX[r * ldS + s_idx] = x;
          }
        }
        __syncthreads();
      }
    }
  }
  if (a.h_out != nullptr)
    for (int i = tid; i < VD_ROWS * R; i += VD_NW * 64) {
      const int r = i / R, c = i - r * R;
      if (b0 + r < B) a.h_out[(size_t)(b0 + r) * R + c] = sH[r * ldR + c];
    }
#ifdef VD_PROF
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0 && a.h_out != nullptr)
    for (int k = 0; k < 7; ++k) a.h_out[k] = (float)(ph[k] / 1000);
#endif
}

size_t vd_lds_bytes(int S, int H, int Z, int R) {
  return sizeof(float) * ((size_t)VD_ROWS * ((S + 4) + 4 * (H + 4) + (R + 4) + (Z + 4) + (VD_CHUNK * VD_F + 4)) + 1024 + 32 + VD_ROWS * VD_CHUNK * 32 + 32 * 16);  // + the op program
}

struct VdPack { size_t enc[3], prior[3], prior_h, phi[4], wih, whh, dec[3], total; };
VdPack vd_pack_layout(int S, int H, int Z, int R) {
  VdPack p;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
  p.enc[0] = take((size_t)H * S); p.enc[1] = take((size_t)H * H); p.enc[2] = take((size_t)H * H);
  p.prior[0] = take((size_t)H * R); p.prior[1] = take((size_t)H * H); p.prior[2] = take((size_t)H * H);
  p.prior_h = take((size_t)2 * Z * H);
  p.phi[0] = take((size_t)H * Z);
  for (int i = 1; i < 4; ++i) p.phi[i] = take((size_t)H * H);
  p.wih = take((size_t)3 * R * 2 * H); p.whh = take((size_t)3 * R * R);
  p.dec[0] = take((size_t)H * (H + R)); p.dec[1] = take((size_t)H * H); p.dec[2] = take((size_t)S * VD_F * H);
  p.total = o;
  return p;
}

}  // namespace
}  // namespace blvm

using namespace blvm;

extern "C" size_t blvm_vrnn_decode_scratch_floats(int S, int H, int Z, int R) {
  if (S <= 0 || H <= 0 || Z <= 0 || R <= 0) return 0;
  return vd_pack_layout(S, H, Z, R).total;
}

extern "C" int blvm_vrnn_decode(const BlvmVrnnDecodeWeights* w, const float* x0, const float* h0, const float* eps, const float* u,
                                const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps, float slope,
                                float log_eps, float* x_out, float* h_out, float* scratch, void* stream_) {
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(w && w->cell && x0 && eps && x_out && scratch, "vrnn_decode: null pointer");
  BLVM_REQUIRE(T >= 0 && B > 0, "vrnn_decode: bad T=%d B=%d", T, B);
  BLVM_REQUIRE(S % 16 == 0 && H % 16 == 0 && Z % 16 == 0 && R % 16 == 0 && S > 0 && H > 0 && Z > 0 && R > 0,
               "vrnn_decode: S, H, Z, R must be positive multiples of 16 (got %d, %d, %d, %d)", S, H, Z, R);
  BLVM_REQUIRE(num_mix == VD_K, "vrnn_decode: the DMoL head has %d components", VD_K);
  BLVM_REQUIRE(14 + 2 * (S / VD_CHUNK) <= 32, "vrnn_decode: frame stacks of more than %d samples are not supported", 9 * VD_CHUNK);
  BLVM_REQUIRE((R / 16 + VD_NW - 1) / VD_NW <= 4, "vrnn_decode: recurrent size %d too large", R);
  BLVM_REQUIRE((u == nullptr) == (v == nullptr), "vrnn_decode: u and v are given together (both NULL: the mode)");
  BLVM_REQUIRE(aligned16(scratch), "vrnn_decode: scratch must be 16-byte aligned");
  const size_t lds = vd_lds_bytes(S, H, Z, R);
  BLVM_REQUIRE(lds <= 160 * 1024, "vrnn_decode: S=%d H=%d Z=%d R=%d need %zu bytes of LDS (> 160 KB)", S, H, Z, R, lds);
  if (T == 0) return BLVM_OK;
  const BlvmVrnnWeights* c = w->cell;
  const VdPack p = vd_pack_layout(S, H, Z, R);
  VDArgs a{};
  int rc;
#define PACK(dst, src, ld, rows, k)                               \
  do {                                                            \
    rc = t16_pack_rows(src, ld, rows, k, scratch + (dst), s);     \
    if (rc) return rc;                                            \
  } while (0)
  PACK(p.enc[0], w->enc_w[0], S, H, S); PACK(p.enc[1], w->enc_w[1], H, H, H); PACK(p.enc[2], w->enc_w[2], H, H, H);
  PACK(p.prior[0], c->prior_w[0], R, H, R); PACK(p.prior[1], c->prior_w[1], H, H, H); PACK(p.prior[2], c->prior_w[2], H, H, H);
  PACK(p.prior_h, c->prior_hw, H, 2 * Z, H);
  PACK(p.phi[0], c->phi_w[0], Z, H, Z);
  for (int i = 1; i < 4; ++i) PACK(p.phi[i], c->phi_w[i], H, H, H);
  PACK(p.wih, c->gru_wih, 2 * H, 3 * R, 2 * H); PACK(p.whh, c->gru_whh, R, 3 * R, R);
  PACK(p.dec[0], w->dec_w[0], H + R, H, H + R); PACK(p.dec[1], w->dec_w[1], H, H, H); PACK(p.dec[2], w->dec_w[2], H, S * VD_F, H);
#undef PACK
  for (int i = 0; i < 3; ++i) {
    a.enc_w[i] = scratch + p.enc[i]; a.enc_b[i] = w->enc_b[i];
    a.prior_w[i] = scratch + p.prior[i]; a.prior_b[i] = c->prior_b[i];
    a.dec_w[i] = scratch + p.dec[i]; a.dec_b[i] = w->dec_b[i];
  }
  for (int i = 0; i < 4; ++i) { a.phi_w[i] = scratch + p.phi[i]; a.phi_b[i] = c->phi_b[i]; }
  a.prior_hw = scratch + p.prior_h; a.prior_hb = c->prior_hb;
  a.wih = scratch + p.wih; a.whh = scratch + p.whh; a.bih = c->gru_bih; a.bhh = c->gru_bhh;
  a.lik_w = w->lik_w; a.lik_b = w->lik_b;
  a.x0 = x0; a.h0 = h0; a.eps = eps; a.u = u; a.v = v; a.x_out = x_out; a.h_out = h_out;
  a.T = T; a.B = B; a.S = S; a.H = H; a.Z = Z; a.R = R;
  a.sd_eps = sd_eps; a.beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps)); a.slope = slope; a.log_eps = log_eps;
  BLVM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vrnn_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(vrnn_decode_kernel, dim3((unsigned)((B + VD_ROWS - 1) / VD_ROWS)), dim3(VD_NW * 64), lds, s, a);
  BLVM_CHECK_LAUNCH("vrnn_decode");
  return BLVM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same sampling loop on the persistent-chain engine (pchain.h / pchain.hip): K1c above keeps 16 utterances on ONE CU (0.37 ms per
// step at any batch size: the fp32 matrix pipe of one CU); here every layer of a step is a link whose 16x16 tiles are dealt over the
// whole chip — 17 descriptors per step, the weights (12.5 MB) stay in the L2s, activations travel as sentinel-polled T16 copies.
// Nothing is kept for a backward pass, but every buffer is still a per-step slab (every word is written once per launch).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace blvm {
namespace {
struct VgBufs {
  size_t X16, E16[2], CAT16, H16, HS, P16[3], GHb, Z16, F16[3], DC16, D16[2], DEC, dummyZ, dummyR, end;
};
VgBufs vg_layout(size_t base, int T, int B, int S, int H, int Z, int R) {
  VgBufs b;
  size_t o = base;
  auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
  const size_t rows = (size_t)((B + 15) / 16) * 16, m = (size_t)T * rows, X = H;
  b.X16 = take((m + rows) * S);
  b.E16[0] = take(m * H); b.E16[1] = take(m * H);
  b.CAT16 = take(m * (X + H));
  b.H16 = take((m + rows) * R);
  b.HS = take((size_t)(T + 1) * B * R);
  for (int i = 0; i < 3; ++i) b.P16[i] = take(m * H);
  b.GHb = take((size_t)T * B * 3 * R);
  b.Z16 = take(m * Z);
  for (int i = 0; i < 3; ++i) b.F16[i] = take(m * H);
  b.DC16 = take(m * (H + R));
  b.D16[0] = take(m * H); b.D16[1] = take(m * H);
  b.DEC = take((size_t)T * B * S * VD_F);
  b.dummyZ = take((size_t)B * Z);
  b.dummyR = take((size_t)B * R);
  b.end = o;
  return b;
}
}  // namespace
}  // namespace blvm

extern "C" size_t blvm_vrnn_generate_scratch_floats(int T, int B, int S, int H, int Z, int R) {
  if (T <= 0 || B <= 0 || S <= 0 || H <= 0 || Z <= 0 || R <= 0) return 0;
  return vg_layout(vd_pack_layout(S, H, Z, R).total, T, B, S, H, Z, R).end;
}

extern "C" int blvm_vrnn_generate(const BlvmVrnnDecodeWeights* w, const float* x0, const float* h0, const float* eps, const float* u,
                                  const float* v, int T, int B, int S, int H, int Z, int R, int num_mix, float sd_eps, float slope,
                                  float log_eps, float* x_out, float* h_out, float* scratch, void* stream_) {
  using namespace pchain;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  BLVM_REQUIRE(w && w->cell && x0 && eps && x_out && scratch, "vrnn_generate: null pointer");
  BLVM_REQUIRE(T >= 0 && B > 0 && B <= kPchainCarveMaxB, "vrnn_generate: bad T=%d B=%d (at most %d utterances)", T, B, kPchainCarveMaxB);
  BLVM_REQUIRE(S % 16 == 0 && H % 16 == 0 && Z % 16 == 0 && R % 16 == 0 && S > 0 && H > 0 && Z > 0 && R > 0,
               "vrnn_generate: S, H, Z, R must be positive multiples of 16 (got %d, %d, %d, %d)", S, H, Z, R);
  BLVM_REQUIRE(num_mix == VD_K, "vrnn_generate: the DMoL head has %d components", VD_K);
  BLVM_REQUIRE((u == nullptr) == (v == nullptr), "vrnn_generate: u and v are given together (both NULL: the mode)");
  BLVM_REQUIRE(aligned16(scratch), "vrnn_generate: scratch must be 16-byte aligned");
  if (T == 0) return BLVM_OK;
  const BlvmVrnnWeights* c = w->cell;
  const VdPack p = vd_pack_layout(S, H, Z, R);
  const VgBufs b = vg_layout(p.total, T, B, S, H, Z, R);
  int rc;
  T16PackScope pack_scope(pchain_bf16(B), s);
#define PACK(dst, src, ld, rows, k)                               \
  do {                                                            \
    rc = t16_pack_rows(src, ld, rows, k, scratch + (dst), s);     \
    if (rc) return rc;                                            \
  } while (0)
  PACK(p.enc[0], w->enc_w[0], S, H, S); PACK(p.enc[1], w->enc_w[1], H, H, H); PACK(p.enc[2], w->enc_w[2], H, H, H);
  PACK(p.prior[0], c->prior_w[0], R, H, R); PACK(p.prior[1], c->prior_w[1], H, H, H); PACK(p.prior[2], c->prior_w[2], H, H, H);
  PACK(p.prior_h, c->prior_hw, H, 2 * Z, H);
  PACK(p.phi[0], c->phi_w[0], Z, H, Z);
  for (int i = 1; i < 4; ++i) PACK(p.phi[i], c->phi_w[i], H, H, H);
  PACK(p.wih, c->gru_wih, 2 * H, 3 * R, 2 * H); PACK(p.whh, c->gru_whh, R, 3 * R, R);
  PACK(p.dec[0], w->dec_w[0], H + R, H, H + R); PACK(p.dec[1], w->dec_w[1], H, H, H); PACK(p.dec[2], w->dec_w[2], H, S * VD_F, H);
#undef PACK
  rc = pack_scope.flush();  // all packs above in one launch
  if (rc) return rc;
  const int rt = (B + 15) / 16, ctS = S / 16, ctH = H / 16, ctZ = Z / 16, ctR = R / 16, X = H, cus = device_cus() & ~7;
  const long rows = (long)rt * 16, xS = rows * S, xH = rows * H, xZ = rows * Z, xR = rows * R, xC = rows * (X + H), xD = rows * (H + R);
  const long sR = (long)B * R, s3R = 3 * sR, sZ = (long)B * Z, sF = (long)B * S * VD_F;
  float* const sc = scratch;
  const float beta = (float)(0.6931471805599453 / (1.0 - (double)sd_eps));
  // ranges: the hidden projection and the wide last decoder layer off to the side of the critical links
  const int r_side = range_for(3 * ctR * rt, std::min(cus / 4, 64));
  const int r_main = range_for(std::max(ctR * rt, ctH * rt), cus - r_side);
  Builder bld;
    bld.p.bf16 = pchain_bf16(B);
  bld.p.S = T; bld.p.B = B; bld.p.xcd = (pchain_tune() & 4) ? 1 : 0; bld.p.lds_products = 4;
  bld.p.prof = pchain_profile_buffer(); bld.p.prof_wg = r_main;
  auto lin = [&](size_t A16, long a_step, size_t W, int K, const float* bias, int ct, int flags, float sl, float* orm, long rm_step, int ldo, size_t o16,
                 long o16_step, int n16, size_t o16b, long o16b_step, int n16b, int wg0, int nwg) -> Desc& {
    Desc& d = bld.add(K_LIN, ct, wg0, nwg, K, flags, 0, T);
    bld.ptr(d, 0, sc + A16, a_step); bld.ptr(d, 1, sc + W); bld.ptr(d, 2, bias); bld.ptr(d, 5, orm, rm_step);
    bld.ptr(d, 6, o16 ? sc + o16 : nullptr, o16_step); bld.ptr(d, 7, o16b ? sc + o16b : nullptr, o16b_step);
    d.ld[3] = ldo; d.n16[0] = n16; d.n16[1] = n16b; d.f[0] = sl;
    return d;
  };
  const int rH = range_for(ctH * rt, r_main);
  // encoder(x_t)
  lin(b.X16, xS, p.enc[0], S, w->enc_b[0], ctH, DF_RELU, slope, nullptr, 0, 0, b.E16[0], xH, ctH, 0, 0, 0, 0, rH);
  lin(b.E16[0], xH, p.enc[1], H, w->enc_b[1], ctH, DF_RELU, slope, nullptr, 0, 0, b.E16[1], xH, ctH, 0, 0, 0, 0, rH);
  lin(b.E16[1], xH, p.enc[2], H, w->enc_b[2], ctH, DF_RELU, slope, nullptr, 0, 0, b.CAT16, xC, (X + H) / 16, 0, 0, 0, 0, rH);
  // prior(h_{t-1}) | hidden projection of the GRU
  lin(b.H16, xR, p.prior[0], R, c->prior_b[0], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.P16[0], xH, ctH, 0, 0, 0, 0, rH);
  lin(b.H16, xR, p.whh, R, c->gru_bhh, 3 * ctR, DF_RM_SC1 | DF_GENTLE | ((pchain_tune() & 16) ? DF_CANARY : 0), 0.f, sc + b.GHb, s3R, 3 * R, 0, 0, 0, 0, 0, 0,
      r_main, r_side);
  const bool seq = linseq_enabled();  // runs of links of one shape as one descriptor (K_LINSEQ)
  if (seq) {
    const SeqLink lp[2] = {{sc + p.prior[1], c->prior_b[1], nullptr, 0, 0, sc + b.P16[1]}, {sc + p.prior[2], c->prior_b[2], nullptr, 0, 0, sc + b.P16[2]}};
    add_linseq(bld, ctH, 0, rH, H, true, false, 0, T, sc + b.P16[0], xH, 2, lp, 0, xH, ctH, 0.f, 0);
  } else {
    lin(b.P16[0], xH, p.prior[1], H, c->prior_b[1], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.P16[1], xH, ctH, 0, 0, 0, 0, rH);
    lin(b.P16[1], xH, p.prior[2], H, c->prior_b[2], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.P16[2], xH, ctH, 0, 0, 0, 0, rH);
  }
  {  // z ~ prior (head in generation mode: the posterior operands are the prior's)
    Desc& d = bld.add(K_HEAD, ctZ, 0, range_for(ctZ * rt, r_main), H, 0, 0, T);
    bld.ptr(d, 0, sc + b.P16[2], xH); bld.ptr(d, 1, sc + b.P16[2], xH); bld.ptr(d, 2, sc + p.prior_h); bld.ptr(d, 3, c->prior_hb);
    bld.ptr(d, 4, sc + p.prior_h); bld.ptr(d, 5, c->prior_hb); bld.ptr(d, 6, eps, sZ);
    for (int k = 7; k <= 12; ++k) bld.ptr(d, k, sc + b.dummyZ);
    bld.ptr(d, 13, nullptr); bld.ptr(d, 14, sc + b.dummyZ); bld.ptr(d, 15, sc + b.Z16, xZ);
    d.ld[3] = Z; d.n16[0] = ctZ; d.i[0] = Z; d.i[1] = 3; d.f[0] = beta; d.f[1] = 1.f / beta; d.f[2] = sd_eps;
  }
  // phi_z(z): the last layer feeds the GRU input cat[enc, phi] and the decoder input cat[phi, h_new]
  if (seq) {
    const int f0 = Z == H ? 0 : 1;  // (the first layer's K is Z)
    if (f0) lin(b.Z16, xZ, p.phi[0], Z, c->phi_b[0], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.F16[0], xH, ctH, 0, 0, 0, 0, rH);
    SeqLink lf[3];
    for (int l = f0; l < 3; ++l) lf[l - f0] = SeqLink{sc + p.phi[l], c->phi_b[l], nullptr, 0, 0, sc + b.F16[l]};
    add_linseq(bld, ctH, 0, rH, H, true, false, 0, T, f0 ? sc + b.F16[0] : sc + b.Z16, f0 ? xH : xZ, 3 - f0, lf, 0, xH, ctH, 0.f, 0);
  } else {
    lin(b.Z16, xZ, p.phi[0], Z, c->phi_b[0], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.F16[0], xH, ctH, 0, 0, 0, 0, rH);
    lin(b.F16[0], xH, p.phi[1], H, c->phi_b[1], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.F16[1], xH, ctH, 0, 0, 0, 0, rH);
    lin(b.F16[1], xH, p.phi[2], H, c->phi_b[2], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.F16[2], xH, ctH, 0, 0, 0, 0, rH);
  }
  lin(b.F16[2], xH, p.phi[3], H, c->phi_b[3], ctH, DF_RELU, 0.f, nullptr, 0, 0, b.CAT16 + (size_t)(X / 16) * 256, xC, (X + H) / 16, b.DC16, xD,
      (H + R) / 16, 0, rH);
  {  // GRU(cat[enc, phi], h_{t-1}) -> h_t: row-major (polled words of the next step), T16 for the next step, T16 into cat[phi, h_t]
    Desc& d = bld.add(K_GRU, ctR, 0, range_for(ctR * rt, r_main), X + H, 0, 0, T);
    bld.ptr(d, 0, sc + b.CAT16, xC); bld.ptr(d, 1, sc + p.wih); bld.ptr(d, 2, nullptr); bld.ptr(d, 3, sc + b.GHb, s3R); bld.ptr(d, 4, sc + b.HS, sR);
    bld.ptr(d, 5, sc + b.HS + sR, sR); bld.ptr(d, 6, sc + b.H16 + xR, xR); bld.ptr(d, 7, sc + b.dummyR); bld.ptr(d, 8, sc + b.dummyR);
    bld.ptr(d, 9, sc + b.dummyR); bld.ptr(d, 10, c->gru_bih); bld.ptr(d, 11, sc + b.DC16 + (size_t)(H / 16) * 256, xD);
    d.ld[0] = R; d.ld[3] = R; d.n16[0] = ctR; d.n16[1] = (H + R) / 16; d.i[0] = R;
  }
  // decoder(cat[phi, h_t]); the last layer (S * F columns) on every workgroup
  lin(b.DC16, xD, p.dec[0], H + R, w->dec_b[0], ctH, DF_RELU, slope, nullptr, 0, 0, b.D16[0], xH, ctH, 0, 0, 0, 0, rH);
  lin(b.D16[0], xH, p.dec[1], H, w->dec_b[1], ctH, DF_RELU, slope, nullptr, 0, 0, b.D16[1], xH, ctH, 0, 0, 0, 0, rH);
  lin(b.D16[1], xH, p.dec[2], H, w->dec_b[2], S * VD_F / 16, DF_RELU | DF_RM_SC1, slope, sc + b.DEC, sF, S * VD_F, 0, 0, 0, 0, 0, 0, 0,
      range_for(S * VD_F / 16 * rt, cus));
  {  // per sample: head Linear -> DMoL draw -> x_{t+1}
    Desc& d = bld.add(K_DMOLS, S / 4, 0, range_for(S / 4 * rt, r_main), 16, 0, 0, T);
    bld.ptr(d, 0, sc + b.DEC, sF); bld.ptr(d, 1, w->lik_w); bld.ptr(d, 2, w->lik_b); bld.ptr(d, 3, u, (long)B * S * VD_K); bld.ptr(d, 4, v, (long)B * S);
    bld.ptr(d, 5, x_out, S); bld.ptr(d, 6, sc + b.X16 + xS, xS);
    d.ld[0] = S * VD_F; d.ld[3] = T * S; d.n16[0] = ctS; d.i[0] = S; d.i[1] = VD_F; d.i[2] = VD_K; d.f[0] = log_eps;
  }
  BLVM_REQUIRE(!bld.overflow, "vrnn_generate: persistent program overflow");
  rc = pchain_ctl(&bld.p.ctl.dev, &bld.p.ctl.host, &bld.p.ctl.epoch);
  if (rc) return rc;
  // sentinel-fill everything the launch polls (all step slabs), then the initial frame stack and state
  BLVM_HIP(pchain_fill_sentinel(sc + b.X16, sizeof(float) * (b.dummyZ - b.X16), s));
  rc = pchain_rows_to_t16(x0, S, B, S, sc + b.X16, s); if (rc) return rc;
  rc = pchain_rows_to_t16(h0, R, B, R, sc + b.H16, s); if (rc) return rc;
  if (h0) BLVM_HIP(hipMemcpyAsync(sc + b.HS, h0, sizeof(float) * (size_t)B * R, hipMemcpyDeviceToDevice, s));
  else BLVM_HIP(hipMemsetAsync(sc + b.HS, 0, sizeof(float) * (size_t)B * R, s));
  rc = pchain_launch(bld.p, s);
  if (rc) return rc;
  if (h_out) BLVM_HIP(hipMemcpyAsync(h_out, sc + b.HS + (size_t)T * sR, sizeof(float) * (size_t)B * R, hipMemcpyDeviceToDevice, s));
  return BLVM_OK;
}
