// common.h — shared host/device helpers for libblvm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "blvm_hip.h"

namespace blvm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);

#define BLVM_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      blvm::set_error(__VA_ARGS__);        \
      return BLVM_EINVAL;                  \
    }                                      \
  } while (0)

#define BLVM_CHECK_LAUNCH(what)                                                     \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      blvm::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));       \
      return BLVM_ELAUNCH;                                                          \
    }                                                                               \
  } while (0)

#define BLVM_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) {                                                        \
      blvm::set_error("%s failed: %s", #call, hipGetErrorString(e__));              \
      return BLVM_ELAUNCH;                                                          \
    }                                                                               \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device math -----------------------------------------------------------------------------------------------
// Accurate libm forms (parity first: ELBO must match the fp32 CPU path to 1e-4 relative over ~1e6 frames).
__device__ __forceinline__ float exp_(float x) { return expf(x); }
__device__ __forceinline__ float log_(float x) { return logf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.nn.functional.softplus(x) (beta=1, threshold=20): x > 20 ? x : log1p(exp(x))
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// softplus with beta (threshold on beta*x, as torch does)
__device__ __forceinline__ float softplus_beta(float x, float beta, float inv_beta) {
  float bx = beta * x;
  return bx > 20.0f ? x : inv_beta * log1pf(expf(bx));
}

// Split-K factor of a weight-gradient GEMM [M,N] += A^T B over K rows: gemm_f32 cuts that form into 64x64 tiles; ask for
// ~3 workgroups per CU, at least 256 k per slice.
inline int gemm_pick_split(int M, int N, int K) {
  const int bm = 64;
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bm - 1) / bm);
  int s = (int)((768 + tiles - 1) / tiles);
  const int kmax = (K + 255) / 256;
  if (s > kmax) s = kmax;
  return s < 1 ? 1 : s;
}

// ---- 16x16 output tile, K split over the NW waves of a workgroup ------------------------------------------------
// acc += A[r0+i][k] * W[c0+j][k] for the k-chunks owned by `wave` (chunk = 16 k, waves interleave chunks).
// A rows >= nrows read as zero.  A, W must be 16-byte aligned with lda, ldw multiples of 4 and K a multiple of 16.
// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+reg.
// Each lane fetches 4 consecutive k as one 16-byte load and feeds them to 4 MFMAs; A and B use the same
// k-permutation, so the sum over k is complete.
//
// WT16: W is in the T16 OPERAND LAYOUT (t16_pack below) instead of row-major.  Row-major, one wave load of the operand
// touches 16 rows x 64 bytes — 16 half-used cache lines — and a workgroup ingests ~38 GB/s from L2; in T16 every
// (16 rows x 16 k) block is stored as the 64 lanes' 16-byte fragments back to back, so the same load is ONE contiguous
// 1 KB read (~150 GB/s per workgroup, tools/cu_ingest.hip).  Block (row tile t, k chunk j) of a [R,K] matrix sits at
// ((t * K/16 + j) * 256) floats, element (rr, 4q+e) of it at (rr + 16 q) * 4 + e: the fragment address of lane
// (rr = lane & 15, q = lane >> 4) is W + c0 * ldw + 16 * k0 + 4 * lane with ldw = the packed matrix's K.
// `mid` hook of the product helpers: runs once, right after the first trip's operand loads are issued (see wave_gemm16_multi)
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};

// keeps the scheduler from hoisting the hook's argument wait (s_waitcnt lgkmcnt) above the operand loads issued before it
template <class Mid>
__device__ __forceinline__ void mid_fence() {
  if constexpr (!__is_same(Mid, NoMid)) __builtin_amdgcn_sched_barrier(0);
}

// U consecutive k-chunks of one wave: all 2U fragment loads first, then the 4U MFMAs in ascending k
template <int U, int STEP, int WS, class Mid>
__device__ __forceinline__ f32x4 gemm16_chunks(const float* __restrict__ ap, const float* __restrict__ wp, int kc, bool aok,
                                                f32x4 acc, bool& pending, Mid& mid) {
  float4 a[U], w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const float4*>(ap + kc + u * STEP);
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const float4*>(wp + (size_t)WS * (kc + u * STEP));
  if (pending) { mid_fence<Mid>(); mid(); pending = false; }
  if (!aok) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, w[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, w[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, w[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, w[u].w, acc, 0, 0, 0);
  }
  return acc;
}

template <int NW, bool WT16 = false, class Mid = NoMid>
__device__ __forceinline__ f32x4 wave_gemm16(const float* __restrict__ A, int lda, int r0, int nrows,
                                              const float* __restrict__ W, int ldw, int c0, int K, int wave,
                                              f32x4 acc, Mid mid = Mid()) {
  bool pending = true;
  constexpr int STEP = NW * 16;
  const int lane = threadIdx.x & 63;
  const int rr = lane & 15, q = lane >> 4;
  const bool aok = (r0 + rr) < nrows;
  const float* ap = A + (size_t)(aok ? r0 + rr : 0) * lda + 4 * q;
  const float* wp = WT16 ? W + (size_t)c0 * ldw + 4 * lane : W + (size_t)(c0 + rr) * ldw + 4 * q;
  constexpr int WS = WT16 ? 16 : 1;  // k stride of the W fragment address
  int kc = wave * 16;
  // 4 chunks per trip keeps 8 x 16-byte loads in flight; left-over chunks are one dependent round trip EACH, so the host picks
  // NW such that a wave owns 4 chunks or 1 (stages.h pick_nw).  Handling 2-3 left-over chunks in one trip here was tried: the
  // extra code costs every link ~0.06 us (VRNN +0.3 ms/step) and only the 3-chunk shapes gain (tools/chain_bench.hip).
  for (; kc + 3 * STEP < K; kc += 4 * STEP) acc = gemm16_chunks<4, STEP, WS>(ap, wp, kc, aok, acc, pending, mid);
  for (; kc < K; kc += STEP) acc = gemm16_chunks<1, STEP, WS>(ap, wp, kc, aok, acc, pending, mid);
  if (pending) mid();
  return acc;
}

// ---- G products of the same K in ONE loop -----------------------------------------------------------------------------
// acc[g] += A[g][r0+i][k] * W[g][c0[g]+j][k], W in T16.  A stage that needs several products (the 4 LSTM gates, the 3 GRU
// gates, both Gaussian heads ...) must not call wave_gemm16 once per product: K is a run-time value, so every call is its own
// loop nest and the loads of product g+1 are issued only after product g's MFMAs — G dependent memory round trips (~1 us each
// from the Infinity Cache) instead of one.  Here all 2G (SAMEA: G+1) fragment loads of a k-chunk are issued together, two
// chunks per trip.  Every acc[g] still sums its chunks in ascending order, element by element, exactly as wave_gemm16 does:
// results are bit-identical.  The MFMAs of the G products are interleaved, so consecutive MFMAs are independent.
//
// `mid` runs once, right after the first trip's operand loads are issued (or at the end for a wave without chunks): a kernel
// whose operand pointers are preloaded SGPRs (stages.h lin1_stage_kernel) puts there the epilogue prefetches whose pointers
// still come by s_load, so the operand loads never wait for the argument fetch.
template <int NW, int G, bool SAMEA, class Mid = NoMid>
__device__ __forceinline__ void wave_gemm16_multi(const float* const (&A)[G], const int (&lda)[G], int r0, int nrows,
                                                  const float* const (&W)[G], const int (&ldw)[G], const int (&c0)[G], int K,
                                                  int wave, f32x4 (&acc)[G], Mid mid = Mid()) {
  bool pending = true;
  constexpr int STEP = NW * 16;
  constexpr int GA = SAMEA ? 1 : G;
  const int lane = threadIdx.x & 63;
  const int rr = lane & 15, q = lane >> 4;
  const bool aok = (r0 + rr) < nrows;
  const float* ap[GA];
  const float* wp[G];
#pragma unroll
  for (int g = 0; g < GA; ++g) ap[g] = A[g] + (size_t)(aok ? r0 + rr : 0) * lda[g] + 4 * q;
#pragma unroll
  for (int g = 0; g < G; ++g) wp[g] = W[g] + (size_t)c0[g] * ldw[g] + 4 * lane;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  int kc = wave * 16;
  for (; kc + STEP < K; kc += 2 * STEP) {
    float4 a0[GA], a1[GA], w0[G], w1[G];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      a0[g] = *reinterpret_cast<const float4*>(ap[g] + kc);
      a1[g] = *reinterpret_cast<const float4*>(ap[g] + kc + STEP);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      w0[g] = *reinterpret_cast<const float4*>(wp[g] + 16 * (size_t)kc);
      w1[g] = *reinterpret_cast<const float4*>(wp[g] + 16 * (size_t)(kc + STEP));
    }
    if (pending) { mid_fence<Mid>(); mid(); pending = false; }
    if (!aok) {
#pragma unroll
      for (int g = 0; g < GA; ++g) a0[g] = a1[g] = zero;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].x, w0[g].x, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].y, w0[g].y, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].z, w0[g].z, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].w, w0[g].w, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[SAMEA ? 0 : g].x, w1[g].x, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[SAMEA ? 0 : g].y, w1[g].y, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[SAMEA ? 0 : g].z, w1[g].z, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[SAMEA ? 0 : g].w, w1[g].w, acc[g], 0, 0, 0);
  }
  for (; kc < K; kc += STEP) {
    float4 a0[GA], w0[G];
#pragma unroll
    for (int g = 0; g < GA; ++g) a0[g] = *reinterpret_cast<const float4*>(ap[g] + kc);
#pragma unroll
    for (int g = 0; g < G; ++g) w0[g] = *reinterpret_cast<const float4*>(wp[g] + 16 * (size_t)kc);
    if (pending) { mid_fence<Mid>(); mid(); pending = false; }
    if (!aok) {
#pragma unroll
      for (int g = 0; g < GA; ++g) a0[g] = zero;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].x, w0[g].x, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].y, w0[g].y, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].z, w0[g].z, acc[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[SAMEA ? 0 : g].w, w0[g].w, acc[g], 0, 0, 0);
  }
  if (pending) mid();
}

// ---- 32x32 output tile (large batches), K split over the NW waves of a workgroup ---------------------------------------
// acc += A[r0+i][k] * W[c0+j][k] with v_mfma_f32_32x32x2_f32: lane (li = lane & 31, lh = lane >> 5) loads 4 consecutive k at
// offset 4*lh of every 8-k chunk as one 16-byte load and feeds them to 4 MFMAs (MFMA c sums k = chunk + c and chunk + 4 + c;
// A and B use the same permutation).  Compared with the 16x16 tile every operand byte fetched from L2 feeds twice the FLOPs:
// at B >= 128 the links are bound by those bytes, not by launch latency.  K must be a multiple of 8.
template <int NW, bool WT16 = false>
__device__ __forceinline__ f32x16 wave_gemm32(const float* __restrict__ A, int lda, int r0, int nrows,
                                               const float* __restrict__ W, int ldw, int c0, int K, int wave, f32x16 acc) {
  constexpr int STEP = NW * 8;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const bool aok = (r0 + li) < nrows;
  const float* ap = A + (size_t)(aok ? r0 + li : 0) * lda + 4 * lh;
  // T16: rows c0 + li live in row tile c0/16 + (li >> 4); k = kc + 4 lh is quad 2 * ((kc >> 3) & 1) + lh of chunk kc >> 4
  const float* wp = WT16 ? W + (size_t)(c0 / 16 + (li >> 4)) * ldw * 16 + ((li & 15) + 16 * lh) * 4 : W + (size_t)(c0 + li) * ldw + 4 * lh;
  constexpr int WS = WT16 ? 16 : 1;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  int kc = wave * 8;
  for (; kc + 3 * STEP < K; kc += 4 * STEP) {
    float4 a0 = *reinterpret_cast<const float4*>(ap + kc);
    float4 a1 = *reinterpret_cast<const float4*>(ap + kc + STEP);
    float4 a2 = *reinterpret_cast<const float4*>(ap + kc + 2 * STEP);
    float4 a3 = *reinterpret_cast<const float4*>(ap + kc + 3 * STEP);
    const float4 w0 = *reinterpret_cast<const float4*>(wp + WS * (kc));
    const float4 w1 = *reinterpret_cast<const float4*>(wp + WS * (kc + STEP));
    const float4 w2 = *reinterpret_cast<const float4*>(wp + WS * (kc + 2 * STEP));
    const float4 w3 = *reinterpret_cast<const float4*>(wp + WS * (kc + 3 * STEP));
    if (!aok) { a0 = a1 = a2 = a3 = zero; }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w1.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, w2.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, w2.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, w2.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, w2.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.x, w3.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.y, w3.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.z, w3.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3.w, w3.w, acc, 0, 0, 0);
  }
  for (; kc < K; kc += STEP) {
    float4 a0 = *reinterpret_cast<const float4*>(ap + kc);
    const float4 w0 = *reinterpret_cast<const float4*>(wp + WS * (kc));
    if (!aok) a0 = zero;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w0.w, acc, 0, 0, 0);
  }
  return acc;
}

// Combine the NW waves' partial 16x16 tiles of G groups through LDS.  `red` must hold G*NW*256 floats.
// After the call threads 0..255 own element (i = tid>>4, j = tid&15) of every group: out[g]; other threads get 0.
template <int G, int NW>
__device__ __forceinline__ void reduce_tiles(const f32x4 (&acc)[G], float* __restrict__ red, float (&out)[G]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // element (row=(lane>>4)*4+reg, col=lane&15) stored at [g][wave][row*16+col]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(g * NW + wave) * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[g][r];
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float s = 0.f;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(g * NW + w) * 256 + tid];
    }
    out[g] = s;
  }
}

}  // namespace blvm
#include <functional>
#include <string>
namespace blvm {
// ---- chain graphs (core.hip): an experiment switch, OFF by default (BLVM_GRAPHS=1 enables) ----------------------------------
// A recurrent sequence is thousands of tiny dependent launches.  What a launch costs (tools/launch_host.hip, graph_chain.hip):
// the GPU's dependent-dispatch floor is 1.53 us; the HOST pays 0.74 us for a launch without arguments but 2.6-3.1 us for one
// with arguments (the runtime writes the argument block into device memory per launch; with HIP_FORCE_DEV_KERNARG=0 the host
// pays 0.9 us and the GPU 3.65 us per kernel fetching arguments from host memory); a captured hipGraph replays a chain of
// trivial kernels at 1.75 us each.  On the VRNN step a chain launch costs the host ~3 us and the GPU ~4 us per link (DESIGN.md,
// "Host or GPU?"), so replaying pre-built graphs looked worth having: run_chain() captures a call and replays it, identified by the bytes
// of ALL its arguments (every pointer, size and flag, the weight / gradient pointer tables included); the first time a key is
// seen the body runs as plain launches, the second time it is captured (on the library's own stream: the caller's may be the
// legacy null stream, which cannot capture), afterwards the graph is replayed between two events on the caller's stream.
// Measured on the real chain (ROCm 7.2): hipGraphLaunch of the 2 260-node forward graph takes 10.9 ms — 4.8 us per node, more
// than launching them — and the step goes from 21.5 to 26.4 ms.  So it stays off.  (Also tried and removed: argument-less twin
// kernels that rebuild their arguments from a device table and a step counter — host enqueue falls to 0.6 us per launch, but
// the table fetch after every kernel boundary and the 10x larger code object put a link at ~5 us on the GPU, 27.3 ms/step.)
// What is left is fewer launches: several links per launch behind an in-kernel barrier.
// The body must only enqueue work on the stream it is given and decide nothing from device data.
struct ChainKey {
  std::string bytes;
  explicit ChainKey(const char* tag) : bytes(tag) {}
  template <class T>
  ChainKey& add(const T& v) {
    bytes.append(reinterpret_cast<const char*>(&v), sizeof(T));
    return *this;
  }
};
int run_chain(const ChainKey& key, hipStream_t user, const std::function<int(hipStream_t)>& body);

// persistent chains (pchain.h): the process-wide control block and a fresh launch epoch (core.hip)
int pchain_ctl(unsigned** dev, unsigned** host_dev, unsigned* epoch);
// largest batch the recurrent sequences run as ONE persistent launch for (0 = never; env BLVM_PCHAIN=0 / BLVM_PCHAIN_MAX_B=n, or
// blvm_pchain_configure).  Beyond it the links are bound by MFMA / operand bytes, not latency, and the 32x32-tile launch-per-link
// kernels fit better.  pchain_waves(): waves per workgroup of the persistent kernels (8 or 16; env BLVM_PCHAIN_NW).
int pchain_max_batch();
int pchain_waves();
int pchain_tune();  // placement bits (env BLVM_PCHAIN_TUNE / blvm_pchain_tune): 4 XCD-aware tile placement, 16 canary polls of deferred tiles
unsigned long long* pchain_profile_buffer();  // diagnostics: null unless blvm_pchain_profile() installed a device buffer

// internal launchers shared between translation units (defined in gemm.hip)
int gemm_f32(int op_a, int op_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
             int ldc, const float* bias, int act, float slope, const float* gate, int ldg, int accumulate,
             int split_k, hipStream_t stream, float* colsum = nullptr);  // colsum (op_a == 1): [M] += sum over k of A[k][:]
int colsum_f32(int M, int N, const float* X, int ldx, float* out, int accumulate, hipStream_t stream);
// Weight gradients of one reduction length as ONE launch (gemm.hip gemm_group_kernel): dW[M,N] += D^T[M,K] Act[K,N] and, with db,
// db[M] += column sums of D, for every job with dW (a job without dW only sums its columns).  Falls back to one gemm_f32 per job
// for bf16 operands, short reductions or more than 20 jobs (env BLVM_WGRAD_GROUP=0: always).
struct WgradJob {
  const float* D; int ldd, M;
  const float* Act; int lda, N;
  float* dW; int ldw;
  float* db;
};
int gemm_wgrad_group(const WgradJob* jobs, int njobs, int K, hipStream_t stream);
int transpose_f32(int M, int N, const float* X, int ldx, float* out, int ldo, hipStream_t stream);
// Operand type of the matrix products (core.hip; blvm_set_operand_dtype / env BLVM_DTYPE=bf16): false = fp32 (the default), true =
// bf16 operands with fp32 accumulation for the persistent chains and K6 — the reference's `--use_amp True` regime
// (experiments/experiment_vrnn_audio.py:219-230).  Everything stored, every epilogue and every reduction stays fp32.
bool operand_bf16();
// While one is alive on this thread, t16_pack() calls on `stream` are collected and launched TOGETHER by flush() (or at the scope's
// end); bf16 = true: the packs are written as bf16 elements (the first half of each dst).
struct T16PackScope {
  T16PackScope(bool bf16, hipStream_t stream);
  ~T16PackScope();
  int flush();
  T16PackScope(const T16PackScope&) = delete;
  T16PackScope& operator=(const T16PackScope&) = delete;
 private:
  bool prev_, prev_active_;
};
// dst = T16 operand layout (see wave_gemm16) of the [R,K] matrix M[r][k] = src[r * rs + k * cs]; R, K multiples of 16.
int t16_pack(const float* src, long rs, long cs, int R, int K, float* dst, hipStream_t stream);
inline int t16_pack_rows(const float* W, int ldw, int R, int K, float* dst, hipStream_t s) { return t16_pack(W, ldw, 1, R, K, dst, s); }
// the transpose of X[:M,:N] (row stride ldx): a [N,M] matrix
inline int t16_pack_transposed(const float* X, int ldx, int M, int N, float* dst, hipStream_t s) { return t16_pack(X, 1, ldx, N, M, dst, s); }

}  // namespace blvm
