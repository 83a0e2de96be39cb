// gemm.hip — K6: fp32 GEMM on v_mfma_f32_32x32x2_f32 with fused bias / activation / activation-derivative epilogue.
//
// Replaces the nn.Linear(+LeakyReLU/ReLU) chains of the reference's encoder/decoder MLPs
// (blvm/models/vrnn.py:487-505, srnn.py:456-474, lstm.py:38-64) and their autograd backward (dgrad / wgrad).
//
// Design (gfx950): 256-thread workgroups = 4 waves in a 2x2 arrangement, block tile BMxBN (128x128 or 64x64),
// BK = 16.  Both operands are staged k-major in LDS (As[k][m], Bs[k][n], row pad 4 floats) so that every MFMA
// operand fetch is a conflict-free ds_read_b32 whatever the global layout (op_a/op_b) was; the transposition is
// done by the global->LDS stage.  Next tile is prefetched into registers while the current one is multiplied.
// f32-in MFMA is an exact fp32 fma chain (guide §3 'FP32-input MFMA'), so results are bit-reproducible except
// for split-K (atomic) accumulation order.
#include <algorithm>
#include <type_traits>

#include <vector>

#include "common.h"

namespace blvm {

namespace {

// LDS operand reads of the NEXT k-pair issued in front of the current pair's MFMAs (2 = pinned with sched_group_barrier: without the
// pins the compiler sinks the reads back next to their use; 0 = the plain loop).  Measured r03: VRNN 15.23 -> 15.18, STCN 32.93 -> 32.51,
// CW-VAE 86.3 -> 85.9 ms/step; the weight-gradient forms +3..4 %, the K = 256 forward forms -1 %.
#ifndef BLVM_GEMM_PIPE
#define BLVM_GEMM_PIPE 2
#endif
#ifndef BLVM_GEMM_BK64
#define BLVM_GEMM_BK64 32
#endif
#ifndef BLVM_GEMM_BK
#define BLVM_GEMM_BK 16
#endif
constexpr int BK = BLVM_GEMM_BK;  // k-tile depth of the 128-wide tiles (and the unit of the host's split arithmetic)
constexpr int tile_bk(int bm, int bn) { return (bm == 64 && bn == 64) ? BLVM_GEMM_BK64 : BK; }

constexpr int PAD = 4;
#ifndef BLVM_GEMM_PAD_T
#define BLVM_GEMM_PAD_T 2
#endif
constexpr int PAD_T = BLVM_GEMM_PAD_T;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* gate;
  int M, N, K, lda, ldb, ldc, ldg;
  int act;
  float slope;
  int accumulate;
  int k_per_split;  // multiple of BK
  int a_vec, b_vec;  // 16-byte vector loads allowed for A / B
  float* colsum;     // op_a == 1 only: [M] += column sums of A over k (the bias gradient of a weight-gradient GEMM), or null
};

// Load 4 consecutive elements p[0..3] with element-wise validity n_valid (0..4).
__device__ __forceinline__ float4 ld4(const float* p, int n_valid, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n_valid >= 4 && vec) {
    v = *reinterpret_cast<const float4*>(p);
  } else {
    if (n_valid > 0) v.x = p[0];
    if (n_valid > 1) v.y = p[1];
    if (n_valid > 2) v.z = p[2];
    if (n_valid > 3) v.w = p[3];
  }
  return v;
}

// one BM x BN output tile over the k range of split `bz` of `nz` (the body of gemm_kernel and of gemm_group_kernel)
template <int BM, int BN, int OPA, int OPB>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz, const int nz) {
  // k-tile depth by tile size: 64 x 64 tiles (8 MFMAs per wave and 16 k, 49 VGPRs at depth 32: still 7 waves per SIMD) run 32 deep --
  // half the barriers per MFMA: weight-gradient forms +4..5 %, WaveNet's K = 96 convs +4 %; the 128-wide tiles would drop from 3 to
  // 2 waves per SIMD at depth 32 (dec L3 forward 195 -> 218 us) and stay at 16.
  constexpr int BK = tile_bk(BM, BN), KV = BK / 4;
  // row pad of the k-major LDS tiles: 4 floats keep the 16-byte stores of an OP == 1 operand aligned; an OP == 0 operand is written
  // by scalar stores whose lanes are 4 k apart (4 rows): with a pad of 2, 4 rows are 8 banks apart and the 32 lanes of a store hit
  // 32 banks (pad 4: 16 banks twice -- SQ_LDS_BANK_CONFLICT was 70 % of the kernel's LDS cycles)
  // (32-deep tiles: 8 lanes per row, a pad of 1 puts 4 rows 4 banks apart)
  constexpr int PT = BLVM_GEMM_PAD_T >= 0 ? (BK == 32 ? (PAD_T + 1) / 2 : PAD_T) : PAD;
  constexpr int LDA_S = BM + (OPA == 0 ? PT : PAD), LDB_S = BN + (OPB == 0 ? PT : PAD);
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave in m / n
  constexpr int A_V = BM * BK / 4 / 256;     // float4 per thread per stage
  constexpr int B_V = BN * BK / 4 / 256;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_V], rb[B_V];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int id = tid + v * 256;
      if (OPA == 0) {  // A[m][k], k contiguous: 4 float4 per row of the tile
        const int m = id / KV, k4 = (id % KV) * 4;
        const int gm = m0 + m, gk = k0 + k4;
        const int nv = (gm < g.M) ? max(0, min(4, kend - gk)) : 0;
        ra[v] = ld4(g.A + (size_t)(gm < g.M ? gm : 0) * g.lda + gk, nv, g.a_vec);
      } else {  // A[k][m], m contiguous
        const int k = id / (BM / 4), m4 = (id % (BM / 4)) * 4;
        const int gm = m0 + m4, gk = k0 + k;
        const int nv = (gk < kend) ? max(0, min(4, g.M - gm)) : 0;
        ra[v] = ld4(g.A + (size_t)(gk < kend ? gk : 0) * g.lda + gm, nv, g.a_vec);
      }
    }
#pragma unroll
    for (int v = 0; v < B_V; ++v) {
      const int id = tid + v * 256;
      if (OPB == 0) {  // B[n][k], k contiguous
        const int n = id / KV, k4 = (id % KV) * 4;
        const int gn = n0 + n, gk = k0 + k4;
        const int nv = (gn < g.N) ? max(0, min(4, kend - gk)) : 0;
        rb[v] = ld4(g.B + (size_t)(gn < g.N ? gn : 0) * g.ldb + gk, nv, g.b_vec);
      } else {  // B[k][n], n contiguous
        const int k = id / (BN / 4), n4 = (id % (BN / 4)) * 4;
        const int gn = n0 + n4, gk = k0 + k;
        const int nv = (gk < kend) ? max(0, min(4, g.N - gn)) : 0;
        rb[v] = ld4(g.B + (size_t)(gk < kend ? gk : 0) * g.ldb + gn, nv, g.b_vec);
      }
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int v = 0; v < A_V; ++v) {
      const int id = tid + v * 256;
      if (OPA == 0) {
        const int m = id / KV, k4 = (id % KV) * 4;
        As[(k4 + 0) * LDA_S + m] = ra[v].x;
        As[(k4 + 1) * LDA_S + m] = ra[v].y;
        As[(k4 + 2) * LDA_S + m] = ra[v].z;
        As[(k4 + 3) * LDA_S + m] = ra[v].w;
      } else {
        const int k = id / (BM / 4), m4 = (id % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(&As[k * LDA_S + m4]) = ra[v];
      }
    }
#pragma unroll
    for (int v = 0; v < B_V; ++v) {
      const int id = tid + v * 256;
      if (OPB == 0) {
        const int n = id / KV, k4 = (id % KV) * 4;
        Bs[(k4 + 0) * LDB_S + n] = rb[v].x;
        Bs[(k4 + 1) * LDB_S + n] = rb[v].y;
        Bs[(k4 + 2) * LDB_S + n] = rb[v].z;
        Bs[(k4 + 3) * LDB_S + n] = rb[v].w;
      } else {
        const int k = id / (BN / 4), n4 = (id % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(&Bs[k * LDB_S + n4]) = rb[v];
      }
    }
  };

  if (kbeg < kend) load_tiles(kbeg);
  const int li = lane & 31, lh = lane >> 5;
  // bias gradient riding on a weight-gradient GEMM: the workgroups of the first column block also sum their A tiles over k
  // (A = the pre-activation gradients [rows, M]; the staged tile is k-major, so thread m reads a conflict-free column)
  const bool do_csum = OPA == 1 && g.colsum != nullptr && bx == 0;
  constexpr int CS = BM <= 64 ? 4 : (BM <= 128 ? 2 : 1);
  float csum = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (k0 + BK < kend) load_tiles(k0 + BK);  // prefetch next tile into registers
    if (do_csum && tid / BM < CS) {  // CS threads share a column: every wave carries the same few extra LDS reads
#pragma unroll
      for (int kk = 0; kk < BK / CS; ++kk) csum += As[(kk * CS + tid / BM) * LDA_S + tid % BM];
    }
#if BLVM_GEMM_PIPE
    // the operands of k-pair s + 1 are requested from LDS BEFORE the MFMAs of k-pair s are issued (two register sets): the LDS
    // round trip hides behind TM x TN matrix instructions instead of standing in front of them
    float a[2][TM], b[2][TN];
    auto lds_ab = [&](int kk, float (&a_)[TM], float (&b_)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a_[i] = As[(kk + lh) * LDA_S + wm * (BM / 2) + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b_[j] = Bs[(kk + lh) * LDB_S + wn * (BN / 2) + j * 32 + li];
    };
    lds_ab(0, a[0], b[0]);
#pragma unroll
    for (int st = 0; st < BK / 2; ++st) {
      if (st + 1 < BK / 2) lds_ab(2 * (st + 1), a[(st + 1) & 1], b[(st + 1) & 1]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][i], b[st & 1][j], acc[i][j], 0, 0, 0);
#if BLVM_GEMM_PIPE >= 2
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // the DS reads of the next pair first ...
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);   // ... then this pair's MFMAs
#endif
    }
#else
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + lh) * LDA_S + wm * (BM / 2) + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + lh) * LDB_S + wn * (BN / 2) + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#endif
  }

  if (do_csum && tid / BM < CS && m0 + tid % BM < g.M) atomicAdd(g.colsum + m0 + tid % BM, csum);
  // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const bool atomic = nz > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + li;
      if (n >= g.N) continue;
      const float bv = (g.bias != nullptr && bz == 0) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        else if (g.act == 2) v = v > 0.f ? v : v * g.slope;
        if (g.gate != nullptr) v *= (g.gate[(size_t)m * g.ldg + n] > 0.f) ? 1.f : g.slope;
        float* cp = g.C + (size_t)m * g.ldc + n;
        if (atomic) atomicAdd(cp, v);
        else if (g.accumulate) *cp += v;
        else *cp = v;
      }
    }
}

template <int BM, int BN, int OPA, int OPB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  gemm_tile<BM, BN, OPA, OPB>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// ---- grouped weight gradients --------------------------------------------------------------------------------------------------
// Up to kMaxGroup problems dW_p (+)= D_p^T Act_p of ONE reduction length K (the rows of a sequence) as one launch: the small ones
// (256 x 256 outputs: 16 tiles) cannot fill 256 CUs on their own without a split so fine that a workgroup's k range is ~20 stages
// and 48 workgroups contend for every output word (tools/gemm_wgrad_sweep.py: 54 TF/s against 81-86 for the 768- and 1920-row
// forms).  Together they are hundreds of tiles: a coarse split, long k ranges, one launch.  Work item = (problem, tile, split):
// first[p] = first work item of problem p (first[n] = all); a workgroup finds its problem by scanning that table.
constexpr int kMaxGroup = 20;
struct GemmGroup {
  GemmArgs p[kMaxGroup];
  int first[kMaxGroup + 1];
  int n, split;
};
__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup gg) {
  const int item = blockIdx.x;
  int p = 0;
  while (p + 1 < gg.n && item >= gg.first[p + 1]) ++p;
  const GemmArgs g = gg.p[p];
  const int local = item - gg.first[p];
  const int tiles_n = (g.N + 63) / 64, tiles = tiles_n * ((g.M + 63) / 64);
  const int bz = local / tiles, t = local - bz * tiles;
  gemm_tile<64, 64, 1, 1>(g, t % tiles_n, t / tiles_n, bz, gg.split);
}

// ---- bf16-operand variant (operand_bf16(): the reference's --use_amp regime) -----------------------------------------------------
// Same tiling, arguments and epilogue; the global -> LDS stage rounds both operands to bf16 (nearest even) and lays them out
// k-group-major — element (m, k) at ((k / 8) * LD + m) * 8 + k % 8 — so that the 8 consecutive k a lane feeds
// v_mfma_f32_32x32x16_bf16 (gfx950's full-rate form) are one ds_read_b128.  BK = 32: two MFMAs per 32x32 tile and staged k-tile
// (16x the fp32 pipe's rate, so this kernel is bound by the operand stream: what it buys is the matrix pipe's time).
// Accumulation and epilogue are fp32.
constexpr int BKB = 32;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2_t){a, b}, b2_t)); }

template <int BM, int BN, int OPA, int OPB>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int KVB = BKB / 4;
  constexpr int A_V = BM * BKB / 4 / 256;  // float4 per thread per stage (2, 4 or 6)
  constexpr int B_V = BN * BKB / 4 / 256;
  static_assert(A_V % 2 == 0 && B_V % 2 == 0, "pairs of k rows");
  __shared__ __attribute__((aligned(16))) unsigned short As[(BKB / 8) * LDA_S * 8];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[(BKB / 8) * LDB_S * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_V], rb[B_V];
  // operand X (rows of length `ext` along m or n): OP == 0: X[row][k], a thread holds 4 consecutive k of one row;
  // OP == 1: X[k][row], a thread holds 4 consecutive rows of k rows 2p and 2p + 1 (items v, v + 1)
  auto load_op = [&](auto op_tag, const float* X, int ldx, int ext, int base, bool vec, int BT, float4* r, int NV, int k0) {
    constexpr int OP = decltype(op_tag)::value;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (OP == 0) {
        const int id = tid + v * 256;
        const int m = id / KVB, k4 = (id % KVB) * 4;
        const int gm = base + m, gk = k0 + k4;
        const int nv = (gm < ext) ? max(0, min(4, kend - gk)) : 0;
        r[v] = ld4(X + (size_t)(gm < ext ? gm : 0) * ldx + gk, nv, vec);
      } else {
        const int id = tid + (v >> 1) * 256;
        const int k = 2 * (id / (BT / 4)) + (v & 1), m4 = (id % (BT / 4)) * 4;
        const int gm = base + m4, gk = k0 + k;
        const int nv = (gk < kend) ? max(0, min(4, ext - gm)) : 0;
        r[v] = ld4(X + (size_t)(gk < kend ? gk : 0) * ldx + gm, nv, vec);
      }
    }
  };
  auto store_op = [&](auto op_tag, unsigned short* S, int LD, int BT, const float4* r, int NV) {
    constexpr int OP = decltype(op_tag)::value;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (OP == 0) {
        const int id = tid + v * 256;
        const int m = id / KVB, k4 = (id % KVB) * 4;
        uint2 q;
        q.x = pk2(r[v].x, r[v].y); q.y = pk2(r[v].z, r[v].w);
        *reinterpret_cast<uint2*>(&S[((k4 >> 3) * LD + m) * 8 + (k4 & 7)]) = q;
      } else if ((v & 1) == 0) {
        const int id = tid + (v >> 1) * 256;
        const int k = 2 * (id / (BT / 4)), m4 = (id % (BT / 4)) * 4;
        unsigned short* d = &S[((k >> 3) * LD + m4) * 8 + (k & 7)];
        *reinterpret_cast<unsigned*>(d + 0) = pk2(r[v].x, r[v + 1].x);
        *reinterpret_cast<unsigned*>(d + 8) = pk2(r[v].y, r[v + 1].y);
        *reinterpret_cast<unsigned*>(d + 16) = pk2(r[v].z, r[v + 1].z);
        *reinterpret_cast<unsigned*>(d + 24) = pk2(r[v].w, r[v + 1].w);
      }
    }
  };
  using OA = std::integral_constant<int, OPA>;
  using OB = std::integral_constant<int, OPB>;
  auto load_tiles = [&](int k0) {
    load_op(OA{}, g.A, g.lda, g.M, m0, g.a_vec != 0, BM, ra, A_V, k0);
    load_op(OB{}, g.B, g.ldb, g.N, n0, g.b_vec != 0, BN, rb, B_V, k0);
  };
  auto store_tiles = [&]() {
    store_op(OA{}, As, LDA_S, BM, ra, A_V);
    store_op(OB{}, Bs, LDB_S, BN, rb, B_V);
  };

  if (kbeg < kend) load_tiles(kbeg);
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = kbeg; k0 < kend; k0 += BKB) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (k0 + BKB < kend) load_tiles(k0 + BKB);
#pragma unroll
    for (int st = 0; st < BKB / 16; ++st) {
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(&As[((2 * st + lh) * LDA_S + wm * (BM / 2) + i * 32 + li) * 8]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(&Bs[((2 * st + lh) * LDB_S + wn * (BN / 2) + j * 32 + li) * 8]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  const bool atomic = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + li;
      if (n >= g.N) continue;
      const float bv = (g.bias != nullptr && blockIdx.z == 0) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        else if (g.act == 2) v = v > 0.f ? v : v * g.slope;
        if (g.gate != nullptr) v *= (g.gate[(size_t)m * g.ldg + n] > 0.f) ? 1.f : g.slope;
        float* cp = g.C + (size_t)m * g.ldc + n;
        if (atomic) atomicAdd(cp, v);
        else if (g.accumulate) *cp += v;
        else *cp = v;
      }
    }
}

template <int BM, int BN>
void launch_gemm(const GemmArgs& g, int op_a, int op_b, dim3 grid, hipStream_t s, bool bf16) {
  if (bf16) {
    if (op_a == 0 && op_b == 0) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 0, 0>), grid, dim3(256), 0, s, g);
    else if (op_a == 0 && op_b == 1) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 0, 1>), grid, dim3(256), 0, s, g);
    else if (op_a == 1 && op_b == 0) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 1, 0>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 1, 1>), grid, dim3(256), 0, s, g);
    return;
  }
  if (op_a == 0 && op_b == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 0>), grid, dim3(256), 0, s, g);
  else if (op_a == 0 && op_b == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 1>), grid, dim3(256), 0, s, g);
  else if (op_a == 1 && op_b == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 0>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 1>), grid, dim3(256), 0, s, g);
}

__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* __restrict__ X, int ldx,
                                                     float* __restrict__ out, int rows_per_block) {
  // block (bx, by): columns bx*64.., rows by*rows_per_block..; 4 row-lanes x 64 columns (scalar fallback: any ldx / alignment)
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += X[(size_t)r * ldx + c];
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// 16-byte loads: a lane owns 4 adjacent columns, a wave 256 columns of one row (1 KiB contiguous); 4 row-lanes (waves)
// nmod: the sums of columns c, c + nmod, c + 2 nmod ... land in out[c % nmod] (a dense [M, n] matrix with n % 4 != 0 read as
// [M / g, n g]: g consecutive rows per "row")
__global__ __launch_bounds__(256) void colsum4_kernel(int M, int N, const float* __restrict__ X, int ldx,
                                                      float* __restrict__ out, int rows_per_block, int nmod) {
  __shared__ float part[4][4][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[rl][0][lane] = s.x; part[rl][1][lane] = s.y; part[rl][2][lane] = s.z; part[rl][3][lane] = s.w;
  __syncthreads();
  if (rl == 0 && c < N) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(out + (c + q) % nmod, part[0][q][lane] + part[1][q][lane] + part[2][q][lane] + part[3][q][lane]);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, const float* y, float slope, float* dz, size_t n4,
                                                      size_t n) {
  // 16 bytes per lane, grid-stride
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = v.x > 0.f ? g.x : g.x * slope;
    o.y = v.y > 0.f ? g.y : g.y * slope;
    o.z = v.z > 0.f ? g.z : g.z * slope;
    o.w = v.w > 0.f ? g.w : g.w * slope;
    reinterpret_cast<float4*>(dz)[i] = o;
  }
  if (blockIdx.x == 0) {
    const size_t i = n4 * 4 + threadIdx.x;
    if (i < n) dz[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
  }
}

__global__ __launch_bounds__(256) void transpose_kernel(int M, int N, const float* __restrict__ X, int ldx,
                                                        float* __restrict__ out, int ldo) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (m0 + i < M && n0 + tx < N) t[i][tx] = X[(size_t)(m0 + i) * ldx + n0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (n0 + i < N && m0 + tx < M) out[(size_t)(n0 + i) * ldo + m0 + tx] = t[tx][i];
}

}  // namespace

int gemm_f32(int op_a, int op_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, int act, float slope, const float* gate, int ldg, int accumulate,
             int split_k, hipStream_t stream, float* colsum) {
  BLVM_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return BLVM_OK;
  BLVM_REQUIRE(A && B && C, "gemm: null operand");
  BLVM_REQUIRE(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
  BLVM_REQUIRE(lda >= (op_a ? M : K) && ldb >= (op_b ? N : K) && ldc >= N, "gemm: leading dimension too small");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.gate = gate;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldg = ldg;
  g.act = act; g.slope = slope; g.accumulate = accumulate;
  g.a_vec = aligned16(A) && (lda % 4 == 0);
  g.b_vec = aligned16(B) && (ldb % 4 == 0);
  if (split_k < 1) split_k = 1;
  bool big = (M >= 128 && N >= 128 && (size_t)((M + 127) / 128) * ((N + 127) / 128) * split_k >= 192);
  // 192-wide tiles for the conv coders' channel counts (192 = 1.5 x 128 would waste a quarter of a 128-tile pair and
  // re-read the other operand): N % 192 == 0 -> 128x192, else M % 192 == 0 -> 192x128
  // Measured on MI355X (tools/gemm_bench.py, tools/gemm_cw.py with BLVM_GEMM_TILE forcing each variant): the 64x64 tile wins
  // for every weight-gradient form (both operands read "transposed": 76 -> 91 TF/s at 768x192x4e5, 71 -> 91 at 1536x512x16000)
  // and for short reductions on narrow outputs (K <= 256 and N <= 256: 43 -> 65 TF/s at K = 96, 53 -> 61 at 256x256); the
  // 128-wide tiles win once N >= 512 or K >= 512 (84 vs 74 TF/s at N = 768, K = 192; 114 vs 95 at 4096^3).
  static const int force_tile = [] { const char* e = getenv("BLVM_GEMM_TILE"); return e ? atoi(e) : 0; }();  // experiments only
  if ((op_a == 1 && op_b == 1) || (K <= 256 && N <= 256)) big = false;
  if (force_tile == 1) big = false;
  const bool n192 = big && N % 192 == 0 && N % 128 != 0 && force_tile != 2;
  const bool m192 = big && !n192 && M % 192 == 0 && M % 128 != 0;
  // (128 x 64 tiles for narrow outputs whose 128 x 128 tiles are fewer than two per CU, measured r03: dec L1 forward 89 -> 85 us, dec L3
  // dgrad unchanged, the VRNN step within noise: not worth a fourth tile shape)
  const int bm = big ? (m192 ? 192 : 128) : 64;
  const int bn = big ? (n192 ? 192 : 128) : 64;
  const bool bf16 = operand_bf16();
  const int bk = bf16 ? BKB : tile_bk(bm, bn);
  if (colsum != nullptr && (bf16 || op_a != 1 || K == 0)) {  // (the bf16 kernel stages ROUNDED operands: the bias gradient stays an fp32 sum)
    const int rc = op_a == 1 ? colsum_f32(K, M, A, lda, colsum, 1, stream) : BLVM_EINVAL;
    if (rc) return rc;
    colsum = nullptr;
  }
  g.colsum = colsum;
  int ksteps = (K + bk - 1) / bk;
  if (split_k > ksteps) split_k = ksteps > 0 ? ksteps : 1;
  g.k_per_split = ((ksteps + split_k - 1) / split_k) * bk;
  if (g.k_per_split == 0) g.k_per_split = bk;
  split_k = (K + g.k_per_split - 1) / g.k_per_split;
  if (split_k < 1) split_k = 1;
  BLVM_REQUIRE(split_k == 1 || (act == 0 && gate == nullptr), "gemm: split-K needs a linear epilogue");
  dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm, split_k);
  BLVM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
  if (split_k > 1 && !accumulate)  // atomic accumulation needs a zeroed destination
    BLVM_HIP(hipMemset2DAsync(C, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)N, (size_t)M, stream));
  if (n192) launch_gemm<128, 192>(g, op_a, op_b, grid, stream, bf16);
  else if (m192) launch_gemm<192, 128>(g, op_a, op_b, grid, stream, bf16);
  else if (big) launch_gemm<128, 128>(g, op_a, op_b, grid, stream, bf16);
  else launch_gemm<64, 64>(g, op_a, op_b, grid, stream, bf16);
  BLVM_CHECK_LAUNCH("gemm_f32");
  return BLVM_OK;
}

int gemm_wgrad_group(const WgradJob* jobs, int njobs, int K, hipStream_t stream) {
  static const int enabled = [] { const char* e = getenv("BLVM_WGRAD_GROUP"); return e ? atoi(e) : 1; }();
  int live = 0;
  for (int i = 0; i < njobs; ++i) live += jobs[i].dW != nullptr;
  if (!enabled || operand_bf16() || live < 2 || live > kMaxGroup || K < 1024) {  // one launch per problem (gemm_f32 picks tile and kernel)
    for (int i = 0; i < njobs; ++i) {
      const WgradJob& j = jobs[i];
      const int rc = j.dW ? gemm_f32(1, 1, j.M, j.N, K, j.D, j.ldd, j.Act, j.lda, j.dW, j.ldw, nullptr, 0, 0.f, nullptr, 0, 1, gemm_pick_split(j.M, j.N, K), stream, j.db)
                          : (j.db ? colsum_f32(K, j.M, j.D, j.ldd, j.db, 1, stream) : BLVM_OK);
      if (rc) return rc;
    }
    return BLVM_OK;
  }
  GemmGroup gg;
  gg.n = 0;
  long tiles = 0;
  for (int i = 0; i < njobs; ++i) {
    const WgradJob& j = jobs[i];
    if (!j.dW) {
      if (j.db) { const int rc = colsum_f32(K, j.M, j.D, j.ldd, j.db, 1, stream); if (rc) return rc; }
      continue;
    }
    BLVM_REQUIRE(j.D && j.Act && j.M > 0 && j.N > 0 && j.ldd >= j.M && j.lda >= j.N && j.ldw >= j.N, "gemm_wgrad_group: bad job %d", i);
    GemmArgs& g = gg.p[gg.n];
    g = GemmArgs{};
    g.A = j.D; g.B = j.Act; g.C = j.dW; g.bias = nullptr; g.gate = nullptr;
    g.M = j.M; g.N = j.N; g.K = K; g.lda = j.ldd; g.ldb = j.lda; g.ldc = j.ldw; g.ldg = 0;
    g.act = 0; g.slope = 0.f; g.accumulate = 1;
    g.a_vec = aligned16(j.D) && (j.ldd % 4 == 0);
    g.b_vec = aligned16(j.Act) && (j.lda % 4 == 0);
    g.colsum = j.db;
    tiles += (long)((j.M + 63) / 64) * ((j.N + 63) / 64);
    ++gg.n;
  }
  // ~6 workgroups per CU, k ranges of at least 32 stages
  constexpr int GBK = tile_bk(64, 64);
  const int ksteps = (K + GBK - 1) / GBK;
  static const int want = [] { const char* e = getenv("BLVM_WGRAD_GROUP_WGS"); return e ? atoi(e) : 1536; }();
  int split = (int)((want + tiles - 1) / tiles);
  split = std::max(1, std::min(split, ksteps / 16));  // k ranges of at least 16 stages (512 rows)
  const int k_per_split = ((ksteps + split - 1) / split) * GBK;
  split = (K + k_per_split - 1) / k_per_split;
  gg.split = split;
  int first = 0;
  for (int p = 0; p < gg.n; ++p) {
    gg.p[p].k_per_split = k_per_split;
    gg.first[p] = first;
    first += ((gg.p[p].M + 63) / 64) * ((gg.p[p].N + 63) / 64) * split;
  }
  gg.first[gg.n] = first;
  hipLaunchKernelGGL(gemm_group_kernel, dim3((unsigned)first), dim3(256), 0, stream, gg);
  BLVM_CHECK_LAUNCH("gemm_wgrad_group");
  return BLVM_OK;
}

int colsum_f32(int M, int N, const float* X, int ldx, float* out, int accumulate, hipStream_t stream) {
  if (N == 0) return BLVM_OK;
  BLVM_REQUIRE(X && out, "colsum: null operand");
  if (!accumulate) BLVM_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, stream));
  if (M == 0) return BLVM_OK;
  int nmod = N;
  if (N % 4 != 0 && ldx == N && aligned16(X)) {  // dense and narrow (the 30-wide DMoL parameter rows): fold g rows into one
    const int g = N % 2 == 0 ? 2 : 4;
    const int tail = M % g;
    if (tail) {  // the last M % g rows through the scalar kernel
      hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, 1), dim3(256), 0, stream, tail, N, X + (size_t)(M - tail) * ldx, ldx, out, 64);
      M -= tail;
    }
    if (M == 0) return BLVM_OK;
    M /= g; N *= g; ldx *= g;
  }
  if (N % 4 == 0 && ldx % 4 == 0 && aligned16(X)) {
    const int col_blocks = (N / 4 + 63) / 64;
    // every row block ends in one float atomic per column: many row blocks on few columns serialise on the same addresses
    // (1 320 blocks on 192 columns took 135 us for a 35-us stream), so cap the row blocks at 128 and let each wave stream longer
    int row_blocks = 1024 / col_blocks;
    if (row_blocks > 128) row_blocks = 128;
    if (row_blocks < 32) row_blocks = 32;
    int rows_per_block = (M + row_blocks - 1) / row_blocks;
    if (rows_per_block < 64) rows_per_block = 64;
    rows_per_block = (rows_per_block + 3) / 4 * 4;
    dim3 grid(col_blocks, (M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum4_kernel, grid, dim3(256), 0, stream, M, N, X, ldx, out, rows_per_block, nmod);
    BLVM_CHECK_LAUNCH("colsum_f32");
    return BLVM_OK;
  }
  int rows_per_block = 64;
  while ((M + rows_per_block - 1) / rows_per_block > 32768) rows_per_block *= 2;
  dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, M, N, X, ldx, out, rows_per_block);
  BLVM_CHECK_LAUNCH("colsum_f32");
  return BLVM_OK;
}

__global__ __launch_bounds__(256) void t16_pack_kernel(const float* __restrict__ src, long rs, long cs, int KB, size_t n,
                                                       float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const size_t blk = i >> 8;
  const int j = (int)(blk % KB);
  const size_t t = blk / KB;
  const size_t r = 16 * t + (lane & 15);
  const int k = 16 * j + 4 * (lane >> 4) + e;
  dst[i] = src[r * rs + (size_t)k * cs];
}

// the same block order with bf16 elements (two per 32-bit word, round to nearest even): the weights of the bf16-operand chains
__global__ __launch_bounds__(256) void t16_pack_bf16_kernel(const float* __restrict__ src, long rs, long cs, int KB, size_t n2,
                                                            unsigned* __restrict__ dst) {
  const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;  // output word = elements 2w, 2w + 1
  if (w >= n2) return;
  const size_t i = 2 * w;
  const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const size_t blk = i >> 8;
  const int j = (int)(blk % KB);
  const size_t t = blk / KB;
  const size_t r = 16 * t + (lane & 15);
  const int k = 16 * j + 4 * (lane >> 4) + e;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const f2 v = {src[r * rs + (size_t)k * cs], src[r * rs + (size_t)(k + 1) * cs]};
  dst[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// Several packs in ONE launch (a recurrent sequence packs 13-15 weight matrices per call, ~4 us of launch each): while a
// T16PackScope is alive on this thread, t16_pack() only records its job; flush() (or the scope's end) launches them together.
constexpr int kPackJobs = 40;
struct PackJob { const float* src; float* dst; long rs, cs; int KB; unsigned n; };  // n: output words of the job
struct PackJobs { PackJob j[kPackJobs]; };
static_assert(sizeof(PackJobs) <= 4000, "kernel argument size");

__global__ __launch_bounds__(256) void t16_pack_jobs_kernel(PackJobs a, int bf16) {
  const PackJob q = a.j[blockIdx.y];
  for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < q.n; w += (size_t)gridDim.x * 256) {
    const size_t i = bf16 ? 2 * w : w;
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int j = (int)(blk % q.KB);
    const size_t t = blk / q.KB;
    const size_t r = 16 * t + (lane & 15);
    const int k = 16 * j + 4 * (lane >> 4) + e;
    if (bf16) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      typedef __bf16 b2 __attribute__((ext_vector_type(2)));
      const f2 v = {q.src[r * q.rs + (size_t)k * q.cs], q.src[r * q.rs + (size_t)(k + 1) * q.cs]};
      reinterpret_cast<unsigned*>(q.dst)[w] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
    } else {
      q.dst[w] = q.src[r * q.rs + (size_t)k * q.cs];
    }
  }
}

namespace {
struct PackState {
  bool active = false, bf16 = false;
  int n = 0;
  hipStream_t stream = nullptr;
  PackJobs jobs;
};
thread_local PackState g_pack;
int pack_flush() {
  PackState& p = g_pack;
  if (p.n == 0) return BLVM_OK;
  unsigned most = 0;
  for (int i = 0; i < p.n; ++i) most = std::max(most, p.jobs.j[i].n);
  const unsigned bx = std::min<unsigned>((most + 255) / 256, 512);
  hipLaunchKernelGGL(t16_pack_jobs_kernel, dim3(bx, (unsigned)p.n), dim3(256), 0, p.stream, p.jobs, p.bf16 ? 1 : 0);
  p.n = 0;
  BLVM_CHECK_LAUNCH("t16_pack_jobs");
  return BLVM_OK;
}
}  // namespace
T16PackScope::T16PackScope(bool bf16, hipStream_t stream) : prev_(g_pack.bf16), prev_active_(g_pack.active) {
  g_pack.bf16 = bf16; g_pack.active = true; g_pack.stream = stream;
}
int T16PackScope::flush() {
  const int rc = pack_flush();
  g_pack.active = false;  // packs requested after the flush run at once again (their consumers may follow immediately)
  return rc;
}
T16PackScope::~T16PackScope() {
  (void)pack_flush();  // (a caller that returned early; an error here resurfaces at the next checked launch)
  g_pack.bf16 = prev_; g_pack.active = prev_active_;
}

int t16_pack(const float* src, long rs, long cs, int R, int K, float* dst, hipStream_t stream) {
  BLVM_REQUIRE(src && dst && R > 0 && K > 0 && R % 16 == 0 && K % 16 == 0 && aligned16(dst), "t16_pack: R=%d, K=%d must be multiples of 16", R, K);
  const size_t n = (size_t)R * K;
  BLVM_REQUIRE(n < (1ull << 31), "t16_pack: matrix too large");
  if (g_pack.active && g_pack.stream == stream) {  // deferred: one launch for all packs of the scope
    if (g_pack.n == kPackJobs) { const int rc = pack_flush(); if (rc) return rc; }
    g_pack.jobs.j[g_pack.n++] = PackJob{src, dst, rs, cs, K / 16, (unsigned)(g_pack.bf16 ? n / 2 : n)};
    return BLVM_OK;
  }
  if (g_pack.bf16) {
    hipLaunchKernelGGL(t16_pack_bf16_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, stream, src, rs, cs, K / 16, n / 2, reinterpret_cast<unsigned*>(dst));
    BLVM_CHECK_LAUNCH("t16_pack_bf16");
    return BLVM_OK;
  }
  hipLaunchKernelGGL(t16_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, rs, cs, K / 16, n, dst);
  BLVM_CHECK_LAUNCH("t16_pack");
  return BLVM_OK;
}

int transpose_f32(int M, int N, const float* X, int ldx, float* out, int ldo, hipStream_t stream) {
  if (M == 0 || N == 0) return BLVM_OK;
  dim3 grid((N + 31) / 32, (M + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, M, N, X, ldx, out, ldo);
  BLVM_CHECK_LAUNCH("transpose_f32");
  return BLVM_OK;
}

}  // namespace blvm

extern "C" int blvm_gemm_f32(int op_a, int op_b, int M, int N, int K, const float* A, int lda, const float* B,
                             int ldb, float* C, int ldc, const float* bias, int act, float slope,
                             const float* gate, int ldg, int accumulate, int split_k, void* stream) {
  return blvm::gemm_f32(op_a, op_b, M, N, K, A, lda, B, ldb, C, ldc, bias, act, slope, gate, ldg, accumulate,
                        split_k, static_cast<hipStream_t>(stream));
}

extern "C" int blvm_wgrad_f32(int N_out, int K_in, int rows, const float* D, int ldd, const float* X, int ldx, float* dW, int lddw, float* db,
                              int split_k, void* stream) {
  using namespace blvm;
  hipStream_t s = static_cast<hipStream_t>(stream);
  BLVM_REQUIRE(N_out >= 0 && K_in >= 0 && rows >= 0, "wgrad: negative dimension");
  if (dW == nullptr) return db ? colsum_f32(rows, N_out, D, ldd, db, 1, s) : BLVM_OK;
  if (split_k < 1) split_k = gemm_pick_split(N_out, K_in, rows);
  return gemm_f32(1, 1, N_out, K_in, rows, D, ldd, X, ldx, dW, lddw, nullptr, 0, 0.f, nullptr, 0, 1, split_k, s, db);
}

extern "C" int blvm_wgrad_group_f32(int n, const int* N_out, const int* K_in, int rows, const float* const* D, const int* ldd, const float* const* X,
                                    const int* ldx, float* const* dW, const int* lddw, float* const* db, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(n >= 0 && rows >= 0 && (n == 0 || (N_out && K_in && D && ldd && X && ldx && dW && lddw && db)), "wgrad_group: bad arguments");
  if (n == 0 || rows == 0) return BLVM_OK;
  std::vector<WgradJob> jobs((size_t)n);
  for (int i = 0; i < n; ++i) {
    BLVM_REQUIRE(N_out[i] > 0 && K_in[i] > 0 && D[i] && (dW[i] == nullptr || X[i]), "wgrad_group: bad job %d", i);
    jobs[(size_t)i] = WgradJob{D[i], ldd[i], N_out[i], X[i], ldx[i], K_in[i], dW[i], lddw[i], db[i]};
  }
  return gemm_wgrad_group(jobs.data(), n, rows, static_cast<hipStream_t>(stream));
}

extern "C" int blvm_act_bwd_f32(const float* dy, const float* y, float slope, float* dz, size_t n, void* stream) {
  using namespace blvm;
  if (n == 0) return BLVM_OK;
  BLVM_REQUIRE(dy && y && dz, "act_bwd: null pointer");
  BLVM_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(dz), "act_bwd: buffers must be 16-byte aligned");
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y,
                     slope, dz, n4, n);
  BLVM_CHECK_LAUNCH("act_bwd_f32");
  return BLVM_OK;
}

extern "C" int blvm_colsum_f32(int M, int N, const float* X, int ldx, float* out, int accumulate, void* stream) {
  return blvm::colsum_f32(M, N, X, ldx, out, accumulate, static_cast<hipStream_t>(stream));
}
