// pchain_probe.hip — what does ONE link of a recurrent chain cost when the whole chain is one persistent launch whose
// workgroups hand 16x16 activation tiles to each other through data-tagged (sentinel-polled) write-through stores
// (benchmarking-lvms_amd/csrc/pchain.h), against the ~4 us of a launch per link?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I benchmarking-lvms_amd/csrc tools/pchain_probe.hip -o tools/pchain_probe
// A chain of L links out_l = relu(out_{l-1} W^T + b), [B,K] x [K,N] with N = K, every link into its own slab.  Checked bit for
// bit against the same chain run as one launch per link.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "pchain.h"
namespace blvm { void set_error(const char*, ...) {} }
using namespace blvm;
using namespace blvm::pchain;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_t16_pack(const float* W, int R, int K, float* dst) {  // [R,K] row-major -> T16
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * K) return;
  const int r = i / K, k = i % K;
  const int t = r / 16, rr = r % 16, j = k / 16, kk = k % 16, q = kk / 4, e = kk % 4;
  dst[((size_t)(t * (K / 16) + j) * 256) + (rr + 16 * q) * 4 + e] = W[i];
}

template <int NW>
__device__ __forceinline__ void lin_tile(const float* A, bool polled, const float* W, const float* bias, float* out, int B, int N,
                                         int K, int r0, int c0, float* red, const Ctl& ctl, unsigned code, bool& dead, bool sc1,
                                         bool local = false) {
  Poll pl{ctl, code, dead, 1, local};
  tile_lin<NW>(A, K, polled, W, K, bias, nullptr, 0, false, nullptr, 0, true, 0.f, out, N, sc1, r0, c0, B, red, pl);
  dead = pl.dead;
}

// the same tile with wall-clock stamps (100 MHz) of wave 0: [0] tile start, [1] operand complete + MFMAs issued, [2] partial tile in LDS
// + barrier + reduction done, [3] store issued, [4] store acknowledged (vmcnt 0)
template <int NW>
__device__ __forceinline__ void lin_tile_prof(const float* A, bool polled, const float* W, const float* bias, float* out, int B, int N,
                                              int K, int r0, int c0, float* red, const Ctl& ctl, unsigned code, bool& dead, bool sc1,
                                              unsigned long long* stamp) {
  const int t = threadIdx.x & 255, wave = threadIdx.x >> 6;
  const int row = r0 + (t >> 4), col = c0 + (t & 15);
  unsigned long long t0 = wall_clock64();
  const float e_bias = bias[col];
  f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
  {
    Poll pl{ctl, code, dead, 1};
    const float* const As[1] = {A};
    const float* const Ws[1] = {W};
    const int la[1] = {K}, cs[1] = {c0};
    mgemm16<NW, 1, 1, MapSame>(As, la, polled, r0, B, Ws, cs, K, acc, pl);
    dead = pl.dead;
  }
  unsigned long long t1 = wall_clock64();
  float v[1];
  reduce_tiles<1, NW>(acc, red, v);
  unsigned long long t2 = wall_clock64();
  if (threadIdx.x < 256 && row < B) {
    float x = v[0] + e_bias;
    x = x > 0.f ? x : 0.f;
    if (sc1) st_sc1(out + (size_t)row * N + col, x);
    else out[(size_t)row * N + col] = x;
  }
  unsigned long long t3 = wall_clock64();
  wait_vm0();
  unsigned long long t4 = wall_clock64();
  if (threadIdx.x == 0) { stamp[0] = t0; stamp[1] = t1; stamp[2] = t2; stamp[3] = t3; stamp[4] = t4; }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_persistent_prof(const float* x0, float* slabs, const float* W, const float* bias, int B, int N,
                                                             int L, int shift, unsigned* ctlw, unsigned long long* stamps) {
  __shared__ float red[2][NW * 256];
  const Ctl ctl{ctlw, ctlw + 4, 1u};
  bool dead = false;
  const int G = gridDim.x, w = blockIdx.x;
  const int rt = (B + 15) / 16, ct = N / 16, ntiles = rt * ct;
  const size_t slab = (size_t)B * N;
  int par = 0;
  for (int l = 0; l < L; ++l) {
    const float* A = l == 0 ? x0 : slabs + (size_t)(l - 1) * slab;
    float* out = slabs + (size_t)l * slab;
    int first = (w - (l * shift) % G + G) % G;
    for (int i = first; i < ntiles; i += G) {
      const int c = i / rt, r = i % rt;
      lin_tile_prof<NW>(A, l > 0, W, bias, out, B, N, N, r * 16, c * 16, red[par], ctl, (unsigned)l, dead, true,
                        stamps + ((size_t)l * ntiles + i) * 5);
      par ^= 1;
    }
  }
}

// persistent: grid = G workgroups; link l's tiles (rt x ct) dealt round-robin starting at workgroup (l * shift) % G
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_persistent(const float* x0, float* slabs, const float* W, const float* bias, int B, int N,
                                                        int L, int shift, unsigned* ctlw) {
  __shared__ float red[2][NW * 256];
  const Ctl ctl{ctlw, ctlw + 4, 1u};
  bool dead = false;
  const int G = gridDim.x, w = blockIdx.x;
  const int rt = (B + 15) / 16, ct = N / 16, ntiles = rt * ct;
  const size_t slab = (size_t)B * N;
  int par = 0;
  for (int l = 0; l < L; ++l) {
    // shift < 0: no dependency between links (every link reads x0, nothing is polled): the compute-only time of a link
    const float* A = (l == 0 || shift < 0) ? x0 : slabs + (size_t)(l - 1) * slab;
    float* out = slabs + (size_t)l * slab;
    int first = shift < 0 ? w : (w - (l * shift) % G + G) % G;
    for (int i = first; i < ntiles; i += G) {
      // all row tiles of one column tile on neighbouring workgroups
      const int c = i / rt, r = i % rt;
      lin_tile<NW>(A, l > 0 && shift >= 0, W, bias, out, B, N, N, r * 16, c * 16, red[par], ctl, (unsigned)l, dead, true);
      par ^= 1;
    }
  }
}

// XCD-grouped: the workgroups of one XCD (one shared L2) run the whole chain of "their" row tiles, so a hand-off never leaves the
// XCD's L2.  Membership is read from the hardware (XCC_ID) and counted at run time; nothing depends on the dispatch order.
// ctlw: [0] abort, [1] code, [2] registered workgroups, [8 + x] members of XCD x.   store_mode 0: sc1 (write-through) stores + sc1
// loads, 1: plain stores + sc1 loads, 2: plain stores + nt loads (L1-bypassing, served by the XCD's own L2).
__device__ __forceinline__ int xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & 15u);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_persistent_xcd(const float* x0, float* slabs, const float* W, const float* bias, int B,
                                                            int N, int L, int shift, int store_mode, unsigned* ctlw) {
  __shared__ float red[2][NW * 256];
  __shared__ int s_rank, s_n;
  const Ctl ctl{ctlw, ctlw + 4, 1u};
  bool dead = false;
  const int G = gridDim.x, xcc = xcc_id();
  if (threadIdx.x == 0) {
    s_rank = (int)__hip_atomic_fetch_add(ctlw + 8 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctlw + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctlw + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
      if (spin_tick(spins, ctl, 0xFFFFu, dead)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    s_n = (int)__hip_atomic_load(ctlw + 8 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int rank = s_rank, n = s_n;
  const int rt = (B + 15) / 16, ct = N / 16;
  const size_t slab = (size_t)B * N;
  int par = 0;
  for (int l = 0; l < L; ++l) {
    const float* A = l == 0 ? x0 : slabs + (size_t)(l - 1) * slab;
    float* out = slabs + (size_t)l * slab;
    for (int r = xcc; r < rt; r += 8) {
      const int first = (rank - (l * shift) % n + n) % n;
      for (int c = first; c < ct; c += n) {
        lin_tile<NW>(A, l > 0, W, bias, out, B, N, N, r * 16, c * 16, red[par], ctl, (unsigned)l, dead, store_mode == 0, store_mode == 2);
        par ^= 1;
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_link(const float* A, float* out, const float* W, const float* bias, int B, int N) {
  __shared__ float red[NW * 256];
  const Ctl ctl{nullptr, nullptr, 1u};  // nothing is polled here
  bool dead = false;
  lin_tile<NW>(A, false, W, bias, out, B, N, N, blockIdx.y * 16, blockIdx.x * 16, red, ctl, 0, dead, false);
}

template <int NW>
int run(int B, int N, int L, int G, int shift, int reps, int xcd_mode = -1) {
  const size_t slab = (size_t)B * N;
  float *x0, *slabs, *ref, *W, *Wt, *bias;
  unsigned* ctlw;
  CK(hipMalloc(&x0, slab * 4)); CK(hipMalloc(&slabs, slab * 4 * L)); CK(hipMalloc(&ref, slab * 4 * 2));
  CK(hipMalloc(&W, (size_t)N * N * 4)); CK(hipMalloc(&Wt, (size_t)N * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&ctlw, 64));
  std::vector<float> h(slab), hw((size_t)N * N), hb(N);
  srand(1);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  // ~orthogonal-ish scale so that activations stay O(1) through thousands of relu layers
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f * 2.45f / sqrtf((float)N);
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-4f;
  CK(hipMemcpy(x0, h.data(), slab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), (size_t)N * N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_t16_pack, dim3((N * N + 255) / 256), dim3(256), 0, 0, W, N, N, Wt);
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemsetAsync(slabs, 0xFF, slab * 4 * L, s));
    CK(hipMemsetAsync(ctlw, 0, 64, s));
    CK(hipEventRecord(e0, s));
    if (xcd_mode < 0) hipLaunchKernelGGL((k_persistent<NW>), dim3(G), dim3(NW * 64), 0, s, x0, slabs, Wt, bias, B, N, L, shift, ctlw);
    else hipLaunchKernelGGL((k_persistent_xcd<NW>), dim3(G), dim3(NW * 64), 0, s, x0, slabs, Wt, bias, B, N, L, shift, xcd_mode, ctlw);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned hc[16];
  CK(hipMemcpy(hc, ctlw, 64, hipMemcpyDeviceToHost));
  // reference: one launch per link, ping-pong
  const dim3 grid(N / 16, (B + 15) / 16);
  CK(hipEventRecord(e0, s));
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL((k_link<NW>), grid, dim3(NW * 64), 0, s, l == 0 ? x0 : ref + ((l - 1) & 1) * slab, ref + (l & 1) * slab, Wt, bias, B, N);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float msl = 0; CK(hipEventElapsedTime(&msl, e0, e1));
  std::vector<float> a(slab), b(slab);
  CK(hipMemcpy(a.data(), slabs + (size_t)(L - 1) * slab, slab * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), ref + ((L - 1) & 1) * slab, slab * 4, hipMemcpyDeviceToHost));
  size_t bad = 0; double mag = 0;
  for (size_t i = 0; i < slab; ++i) { bad += a[i] != b[i]; mag += fabs(b[i]); }
  if (shift < 0) bad = 0;  // compute-only run: results are not the chain's
  if (xcd_mode >= 0) printf("[xcd groups %u %u %u %u %u %u %u %u, %s stores] ", hc[8], hc[9], hc[10], hc[11], hc[12], hc[13], hc[14], hc[15], xcd_mode == 2 ? "plain + nt loads" : (xcd_mode ? "plain" : "sc1"));
  printf("B=%3d N=K=%4d L=%4d G=%3d NW=%2d shift=%2d: persistent %6.3f us/link | launches %6.3f us/link | mismatches %zu/%zu (mean|x| %.3g) aborted launches=%u code=%u\n",
         B, N, L, G, NW, shift, best * 1e3 / L, msl * 1e3 / L, bad, slab, mag / slab, hc[4], hc[5]);
  hipFree(x0); hipFree(slabs); hipFree(ref); hipFree(W); hipFree(Wt); hipFree(bias); hipFree(ctlw);
  return 0;
}

template <int NW>
int run_prof(int B, int N, int L, int G, int shift) {
  const size_t slab = (size_t)B * N;
  const int ntiles = ((B + 15) / 16) * (N / 16);
  float *x0, *slabs, *W, *Wt, *bias;
  unsigned* ctlw;
  unsigned long long* stamps;
  CK(hipMalloc(&x0, slab * 4)); CK(hipMalloc(&slabs, slab * 4 * L));
  CK(hipMalloc(&W, (size_t)N * N * 4)); CK(hipMalloc(&Wt, (size_t)N * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&ctlw, 64));
  CK(hipMalloc(&stamps, (size_t)L * ntiles * 5 * 8));
  std::vector<float> h(slab), hw((size_t)N * N), hb(N);
  srand(1);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f * 2.45f / sqrtf((float)N);
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-4f;
  CK(hipMemcpy(x0, h.data(), slab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), (size_t)N * N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_t16_pack, dim3((N * N + 255) / 256), dim3(256), 0, 0, W, N, N, Wt);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(slabs, 0xFF, slab * 4 * L));
    CK(hipMemset(ctlw, 0, 64));
    hipLaunchKernelGGL((k_persistent_prof<NW>), dim3(G), dim3(NW * 64), 0, 0, x0, slabs, Wt, bias, B, N, L, shift, ctlw, stamps);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> st((size_t)L * ntiles * 5);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  // per link: the last producer's store-issue / store-ack time of link l-1 against every consumer's operand-complete time of link l
  double wait_w = 0, poll = 0, red = 0, sti = 0, ack = 0, hand_issue = 0, hand_ack = 0, link = 0;
  int n = 0;
  for (int l = 100; l < L; ++l) {
    unsigned long long last_issue = 0, last_ack = 0, last_issue_prev = 0;
    for (int i = 0; i < ntiles; ++i) {
      const unsigned long long* p = &st[((size_t)(l - 1) * ntiles + i) * 5];
      if (p[3] > last_issue) last_issue = p[3];
      if (p[4] > last_ack) last_ack = p[4];
      const unsigned long long* pp = &st[((size_t)(l - 2) * ntiles + i) * 5];
      if (pp[3] > last_issue_prev) last_issue_prev = pp[3];
    }
    link += (double)(last_issue - last_issue_prev);
    for (int i = 0; i < ntiles; ++i) {
      const unsigned long long* c = &st[((size_t)l * ntiles + i) * 5];
      poll += (double)(c[1] - c[0]); red += (double)(c[2] - c[1]); sti += (double)(c[3] - c[2]); ack += (double)(c[4] - c[3]);
      hand_issue += (double)((long long)(c[1] - last_issue)); hand_ack += (double)((long long)(c[1] - last_ack));
      ++n;
    }
  }
  const double u = 0.01;  // us per tick
  printf("PROF B=%d N=K=%d G=%d NW=%d shift=%d: link period %.3f us | tile: start->operands %.3f, ->reduced %.3f, ->store issued %.3f, ->store acked %.3f | "
         "last producer's store ISSUE -> consumer operands complete %.3f us, last producer's store ACK -> same %.3f us\n",
         B, N, G, NW, shift, link * u / (L - 100), poll * u / n, red * u / n, sti * u / n, ack * u / n, hand_issue * u / n, hand_ack * u / n);
  return 0;
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 2000;
  if (argc > 2 && atoi(argv[2]) == 3) {
    for (int B : {8, 64})
      for (int N : {256, 512})
        for (int shift : {0, 5}) {
          if (run_prof<16>(B, N, L, 256, shift)) return 1;
          if (run_prof<8>(B, N, L, 256, shift)) return 1;
        }
    return 0;
  }
  if (argc > 2 && atoi(argv[2]) == 1) {  // XCD-grouped variants
    for (int B : {8, 64, 128}) {
      for (int N : {256, 512}) {
        for (int mode : {0, 1, 2}) {
          if (run<8>(B, N, L, 256, 0, 3, mode)) return 1;
          if (run<16>(B, N, L, 256, 0, 3, mode)) return 1;
        }
        if (run<16>(B, N, L, 256, 5, 3, 2)) return 1;
      }
    }
    return 0;
  }
  if (argc > 2 && atoi(argv[2]) == 2) {  // compute-only (no dependency) time per link
    for (int B : {8, 64})
      for (int N : {256, 512}) {
        if (run<8>(B, N, L, 256, -1, 3)) return 1;
        if (run<16>(B, N, L, 256, -1, 3)) return 1;
      }
    return 0;
  }
  for (int B : {8, 64}) {
    for (int N : {256, 512}) {
      for (int G : {64, 128, 256}) {
        if (run<4>(B, N, L, G, 0, 3)) return 1;
        if (run<8>(B, N, L, G, 0, 3)) return 1;
        if (run<16>(B, N, L, G, 0, 3)) return 1;
      }
      if (run<8>(B, N, L, 256, 7, 3)) return 1;
      if (run<8>(B, N, L, 256, 64, 3)) return 1;
    }
  }
  return 0;
}
