// chain_bench.hip — diagnostic (not part of the product): what does ONE link of a dependent launch chain cost on
// MI355X, and where does it go?  Build: hipcc --offload-arch=gfx950 -O3 -I include -I benchmarking-lvms_amd/csrc
// tools/chain_bench.hip -o gpurun_out/chain_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "common.h"
namespace blvm { void set_error(const char*, ...) {} }
using namespace blvm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

struct Big { const float* a[3]; int l[3]; const float* w[3]; int lw[3]; const float* b[3]; const float* ad[3]; int la[3];
             const float* g[3]; int lg[3]; float* o[3]; int lo[3]; int tiles[3]; int relu[3]; int nseg, B, K; };
__global__ void k_bigarg(Big a) {
  int s = blockIdx.x >= a.tiles[0] ? 1 : 0;
  float* o = s == 0 ? a.o[0] : a.o[1];
  if (threadIdx.x == 0) o[blockIdx.x] = (float)(s == 0 ? a.l[0] : a.l[1]);
}

// single-segment packed linear stage (small kernarg, all epilogue operands prefetched unconditionally)
struct Small { const float* A; const float* W; const float* bias; const float* add; float* out;
               int lda, ldw, ldadd, ldo, B, K, a_split, a_off, add_c0, add_c1, relu_c1; };
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_small(Small a) {
  __shared__ float red[NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const int rowc = row < a.B ? row : r0;
  const float e_bias = a.bias[col];
  const bool has_add = c0 >= a.add_c0 && c0 < a.add_c1;
  const float e_add = a.add[(size_t)rowc * a.ldadd + (has_add ? col - a.add_c0 : 0)];
  const float* A = a.A + (c0 >= a.a_split ? a.a_off : 0);
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW>(A, a.lda, r0, a.B, a.W, a.ldw, c0, a.K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (threadIdx.x >= 256 || row >= a.B) return;
  float x = v[0] + e_bias + (has_add ? e_add : 0.f);
  if (c0 < a.relu_c1) x = x > 0.f ? x : 0.f;
  a.out[(size_t)row * a.ldo + col] = x;
}

// XCD-aware tile order: consecutive linear block ids are dealt round-robin over the 8 XCDs, so give every XCD its own
// set of column tiles (all row tiles of a column tile on the same XCD): each XCD's L2 then fetches 1/8 of W.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_small_xcd(Small a) {
  __shared__ float red[NW * 256];
  const int L = blockIdx.x + gridDim.x * blockIdx.y, nrt = gridDim.y;
  const int ct = (L & 7) + 8 * ((L >> 3) / nrt), rt = (L >> 3) % nrt;
  const int r0 = rt * 16, c0 = ct * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const int rowc = row < a.B ? row : r0;
  const float e_bias = a.bias[col];
  const bool has_add = c0 >= a.add_c0 && c0 < a.add_c1;
  const float e_add = a.add[(size_t)rowc * a.ldadd + (has_add ? col - a.add_c0 : 0)];
  const float* A = a.A + (c0 >= a.a_split ? a.a_off : 0);
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW>(A, a.lda, r0, a.B, a.W, a.ldw, c0, a.K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (threadIdx.x >= 256 || row >= a.B) return;
  float x = v[0] + e_bias + (has_add ? e_add : 0.f);
  if (c0 < a.relu_c1) x = x > 0.f ? x : 0.f;
  a.out[(size_t)row * a.ldo + col] = x;
}

// scalar arguments (candidates for SGPR kernarg preload)
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_scalar(const float* A, const float* W, const float* bias, float* out,
                                                    int lda, int ldw, int ldo, int B, int K, int relu_c1) {
  __shared__ float red[NW * 256];
  const int L = blockIdx.x + gridDim.x * blockIdx.y, nrt = gridDim.y;
  const int ct = (L & 7) + 8 * ((L >> 3) / nrt), rt = (L >> 3) % nrt;
  const int r0 = rt * 16, c0 = ct * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const float e_bias = bias[col];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW>(A, lda, r0, B, W, ldw, c0, K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (threadIdx.x >= 256 || row >= B) return;
  float x = v[0] + e_bias;
  if (c0 < relu_c1) x = x > 0.f ? x : 0.f;
  out[(size_t)row * ldo + col] = x;
}

// same but WITHOUT any epilogue loads (pure GEMM + store): isolates the epilogue round trip
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_small_noepi(Small a) {
  __shared__ float red[NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW>(a.A, a.lda, r0, a.B, a.W, a.ldw, c0, a.K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (threadIdx.x >= 256 || row >= a.B) return;
  a.out[(size_t)row * a.ldo + col] = v[0] > 0.f ? v[0] : 0.f;
}


// output-store experiment: MODE 0 plain store, 1 nontemporal store, 2 agent-scope relaxed atomic store (sc1 write-through),
// 3 no store at all (what the store + the kernel-end release of dirty L2 lines cost together), 4 = 16-byte stores by 64 threads
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k_store(Small a) {
  __shared__ float red[NW * 256];
  const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int t = threadIdx.x & 255;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  const float e_bias = a.bias[col];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  acc[0] = wave_gemm16<NW>(a.A, a.lda, r0, a.B, a.W, a.ldw, c0, a.K, threadIdx.x >> 6, acc[0]);
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  if (threadIdx.x >= 256 || row >= a.B) return;
  float x = v[0] + e_bias;
  x = x > 0.f ? x : 0.f;
  float* o = a.out + (size_t)row * a.ldo + col;
  if (MODE == 0) *o = x;
  else if (MODE == 1) __builtin_nontemporal_store(x, o);
  else if (MODE == 2) __hip_atomic_store(o, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 3) { if (x == 12345.f) *o = x; }
}


// pair-tile experiment: one workgroup of 2*NW waves owns a 16x32 output tile (two adjacent 16x16 tiles, NW waves each), so
// the A rows are fetched by half as many workgroups and every output row segment is a full 128-byte line.
// Measured: 4.2 us per link against 3.3 us for the 16x16 tiles (N=512, K=256) — fewer, fatter workgroups lose.
template <int NW>
__global__ __launch_bounds__(NW * 128) void k_pair(Small a) {
  __shared__ float red[2 * NW * 256];
  const int wave = threadIdx.x >> 6, half = wave / NW, wv = wave % NW, lane = threadIdx.x & 63;
  const int r0 = blockIdx.y * 16, c0 = (2 * blockIdx.x + half) * 16;
  const int tid = threadIdx.x, orow = tid >> 5, occ = tid & 31;
  const float e_bias = a.bias[2 * blockIdx.x * 16 + occ];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = wave_gemm16<NW>(a.A, a.lda, r0, a.B, a.W, a.ldw, c0, a.K, wv, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(half * NW + wv) * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();
  if (tid >= 512 || r0 + orow >= a.B) return;
  float x = e_bias;
#pragma unroll
  for (int w = 0; w < NW; ++w) x += red[((occ >> 4) * NW + w) * 256 + orow * 16 + (occ & 15)];
  x = x > 0.f ? x : 0.f;
  a.out[(size_t)(r0 + orow) * a.ldo + 2 * blockIdx.x * 16 + occ] = x;
}

int main() {
  const int B = 64, N = 512, K = 256, REP = 2000;
  float *act[2], *W, *bias, *add;
  CK(hipMalloc(&act[0], sizeof(float) * B * 2048)); CK(hipMalloc(&act[1], sizeof(float) * B * 2048));
  CK(hipMalloc(&W, sizeof(float) * 2048 * 1536)); CK(hipMalloc(&bias, sizeof(float) * 2048)); CK(hipMalloc(&add, sizeof(float) * B * 2048));
  CK(hipMemset(act[0], 0, sizeof(float) * B * 2048)); CK(hipMemset(act[1], 0, sizeof(float) * B * 2048));
  CK(hipMemset(W, 0, sizeof(float) * 2048 * 1536)); CK(hipMemset(bias, 0, sizeof(float) * 2048)); CK(hipMemset(add, 0, sizeof(float) * B * 2048));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 50; ++i) launch(i);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < REP; ++i) launch(i);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %7.3f us/launch\n", name, ms * 1e3 / REP);
  };
  timeit("empty kernel, 128 WG x 256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(128), dim3(256), 0, s); });
  timeit("empty kernel, 128 WG x 1024", [&](int) { hipLaunchKernelGGL(k_empty, dim3(128), dim3(1024), 0, s); });
  timeit("touch kernel (1 ptr arg), 128 WG x 256", [&](int i) { hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, s, act[i & 1]); });

  for (int kk : {256}) {
    Small a{}; a.W = W; a.bias = bias; a.add = add; a.lda = 2048; a.ldw = kk; a.ldadd = 2048; a.ldo = 2048; a.B = B; a.K = kk;
    for (int rep = 0; rep < 2; ++rep) {
      timeit("store mode 0 plain        N=512 K=256 NW=4", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_store<4, 0>), dim3(N / 16, 4), dim3(256), 0, s, a); });
      timeit("store mode 1 nontemporal  N=512 K=256 NW=4", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_store<4, 1>), dim3(N / 16, 4), dim3(256), 0, s, a); });
      timeit("store mode 2 agent atomic N=512 K=256 NW=4", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_store<4, 2>), dim3(N / 16, 4), dim3(256), 0, s, a); });
      timeit("store mode 3 no store     N=512 K=256 NW=4", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_store<4, 3>), dim3(N / 16, 4), dim3(256), 0, s, a); });
    }

    for (int rep = 0; rep < 2; ++rep) {
      timeit("pair tile 16x32, 8 waves  N=512 K=256 NW=4", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_pair<4>), dim3(N / 32, 4), dim3(512), 0, s, a); });
      timeit("pair tile 16x32, 16 waves N=512 K=256 NW=8", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_pair<8>), dim3(N / 32, 4), dim3(1024), 0, s, a); });
      timeit("store mode 0 plain        N=512 K=256 NW=8", [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1]; hipLaunchKernelGGL((k_store<8, 0>), dim3(N / 16, 4), dim3(512), 0, s, a); });
    }
  }
  Big b{}; b.tiles[0] = 64; b.o[0] = act[0]; b.o[1] = act[1]; b.l[0] = 1; b.l[1] = 2;
  timeit("big kernarg (280 B) + select, 128 WG x 256", [&](int) { hipLaunchKernelGGL(k_bigarg, dim3(128), dim3(256), 0, s, b); });
  for (int kk : {192, 256, 512, 768, 1536}) {
    for (int nw : {4, 8, 16}) {
      Small a{}; a.W = W; a.bias = bias; a.add = add; a.lda = 2048; a.ldw = kk; a.ldadd = 2048; a.ldo = 2048; a.B = B; a.K = kk;
      a.a_split = 256; a.a_off = 0; a.add_c0 = 256; a.add_c1 = 512; a.relu_c1 = 512;
      char nm[128];
      snprintf(nm, sizeof nm, "small-arg lin stage N=%d K=%d NW=%d (epilogue prefetched)", N, kk, nw);
      timeit(nm, [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1];
        if (nw == 4) hipLaunchKernelGGL((k_small<4>), dim3(N / 16, 4), dim3(256), 0, s, a);
        else if (nw == 8) hipLaunchKernelGGL((k_small<8>), dim3(N / 16, 4), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_small<16>), dim3(N / 16, 4), dim3(1024), 0, s, a); });
      snprintf(nm, sizeof nm, "   same, no epilogue loads           K=%d NW=%d", kk, nw);
      timeit(nm, [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1];
        if (nw == 4) hipLaunchKernelGGL((k_small_noepi<4>), dim3(N / 16, 4), dim3(256), 0, s, a);
        else if (nw == 8) hipLaunchKernelGGL((k_small_noepi<8>), dim3(N / 16, 4), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_small_noepi<16>), dim3(N / 16, 4), dim3(1024), 0, s, a); });
    }
  }
  for (int kk : {256, 512, 1536}) {
    for (int nn : {256, 512, 2048}) {
      Small a{}; a.W = W; a.bias = bias; a.add = add; a.lda = 2048; a.ldw = kk; a.ldadd = 2048; a.ldo = 2048; a.B = B; a.K = kk;
      a.a_split = 4096; a.add_c0 = 256; a.add_c1 = 512; a.relu_c1 = 512;
      char nm[128];
      snprintf(nm, sizeof nm, "XCD-aware small-arg N=%d K=%d NW=8", nn, kk);
      timeit(nm, [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1];
        hipLaunchKernelGGL((k_small_xcd<8>), dim3(nn / 16, 4), dim3(512), 0, s, a); });
      snprintf(nm, sizeof nm, "   plain order         N=%d K=%d NW=8", nn, kk);
      timeit(nm, [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1];
        hipLaunchKernelGGL((k_small<8>), dim3(nn / 16, 4), dim3(512), 0, s, a); });
      snprintf(nm, sizeof nm, "   XCD-aware, scalar args N=%d K=%d NW=8", nn, kk);
      timeit(nm, [&](int i) {
        hipLaunchKernelGGL((k_scalar<8>), dim3(nn / 16, 4), dim3(512), 0, s, (const float*)act[i & 1], (const float*)W, (const float*)bias, act[(i + 1) & 1], 2048, kk, 2048, B, kk, 512); });
    }
  }
  // wide stage like F1: N=2048
  for (int nw : {4, 8}) {
    Small a{}; a.W = W; a.bias = bias; a.add = add; a.lda = 2048; a.ldw = 512; a.ldadd = 2048; a.ldo = 2048; a.B = B; a.K = 512;
    a.a_split = 4096; a.add_c0 = 256; a.add_c1 = 512; a.relu_c1 = 512;
    char nm[128]; snprintf(nm, sizeof nm, "small-arg lin stage N=2048 K=512 NW=%d", nw);
    timeit(nm, [&](int i) { a.A = act[i & 1]; a.out = act[(i + 1) & 1];
      if (nw == 4) hipLaunchKernelGGL((k_small<4>), dim3(128, 4), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_small<8>), dim3(128, 4), dim3(512), 0, s, a); });
  }
  return 0;
}
