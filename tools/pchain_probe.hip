// pchain_probe.hip — what does ONE link of a recurrent chain cost when the whole chain is one persistent launch whose workgroups
// hand 16x16 activation tiles to each other through data-tagged (sentinel-polled) write-through stores
// (benchmarking-lvms_amd/csrc/pchain.h), against a launch per link?  The bare tile loop: no program, no descriptors — the floor
// the engine (csrc/pchain.hip, blvm_pchain_chain_probe) is measured against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I benchmarking-lvms_amd/csrc tools/pchain_probe.hip -o tools/pchain_probe
// A chain of L links out_l = relu(out_{l-1} W^T + b), [B,K] x [K,N] with N = K, every link into its own slab (row-major copy +
// T16 operand copy for the next link).  Checked bit for bit against the same chain run as one launch per link.
// Measured with earlier versions of this probe and removed (numbers in tools/README.md): workgroups grouped by XCD (HW_REG_XCC_ID)
// exchanging through the XCD's own L2 with plain stores + sc1 / nt loads — no faster than write-through stores across the chip;
// two polls in flight per wave; in-kernel wall-clock stamps per tile phase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "pchain.h"
namespace blvm {
void set_error(const char*, ...) {}
int pchain_max_batch() { return 128; }
}  // namespace blvm
using namespace blvm;
using namespace blvm::pchain;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_t16_pack(const float* W, int R, int K, float* dst) {  // [R,K] row-major -> T16 (rows beyond R untouched)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * K) return;
  const int r = i / K, k = i % K;
  dst[((size_t)((r / 16) * (K / 16) + k / 16) * 256) + ((r % 16) + 16 * ((k % 16) / 4)) * 4 + (k % 4)] = W[i];
}

// persistent: grid = G workgroups; link l's tiles (rt x ct) dealt round-robin starting at workgroup (l * shift) % G.
// shift < 0: no dependency between links (every link reads x0, nothing waits): the compute-only time of a link
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_persistent(const float* x0_16, float* slabs, float* slabs16, const float* W, const float* bias, int B,
                                                        int N, int L, int shift, unsigned* ctlw) {
  __shared__ float red[2][NW * 256];
  Poll pl{Ctl{ctlw, ctlw + 4, 1u}, 0u, false, 1};
  const int G = gridDim.x, w = blockIdx.x;
  const int rt = (B + 15) / 16, ct = N / 16, ntiles = rt * ct;
  const size_t slab = (size_t)B * N, slab16 = (size_t)rt * 16 * N;
  int par = 0;
  for (int l = 0; l < L; ++l) {
    const float* A = (l == 0 || shift == -1) ? x0_16 : slabs16 + (size_t)(l - 1) * slab16;
    const Out o = out_both(slabs + (size_t)l * slab, N, slabs16 + (size_t)l * slab16, ct);
    pl.code = (unsigned)l;
    if (shift == -2) {  // every row tile's chain on ONE XCD (workgroup w sits on XCD w % 8): row tile w % 8, column tile w / 8
      const int r = w & 7, c = w >> 3;
      if (r < rt && c < ct) {
        tile_lin<NW>(A, 0, true, W, N, bias, nullptr, 0, false, nullptr, 0, true, 0.f, o, r * 16, c * 16, B, red[par], pl);
        par ^= 1;
      }
      continue;
    }
    const int first = shift < 0 ? w : (w - (l * shift) % G + G) % G;
    for (int i = first; i < ntiles; i += G) {
      tile_lin<NW>(A, 0, true, W, N, bias, nullptr, 0, false, nullptr, 0, true, 0.f, o, (i % rt) * 16, (i / rt) * 16, B, red[par], pl);
      par ^= 1;
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_link(const float* A16, float* out, float* out16, const float* W, const float* bias, int B, int N) {
  __shared__ float red[NW * 256];
  Poll pl{Ctl{nullptr, nullptr, 1u}, 0u, false, 1};  // the data is complete: the first poll succeeds
  tile_lin<NW>(A16, 0, true, W, N, bias, nullptr, 0, false, nullptr, 0, true, 0.f, out_both(out, N, out16, N / 16), blockIdx.y * 16, blockIdx.x * 16, B, red, pl);
}

template <int NW>
int run(int B, int N, int L, int G, int shift, int reps) {
  const size_t slab = (size_t)B * N, slab16 = (size_t)((B + 15) / 16) * 16 * N;
  float *x0, *x0_16, *slabs, *slabs16, *ref, *ref16, *W, *Wt, *bias;
  unsigned* ctlw;
  CK(hipMalloc(&x0, slab * 4)); CK(hipMalloc(&x0_16, slab16 * 4)); CK(hipMalloc(&slabs, slab * 4 * L)); CK(hipMalloc(&slabs16, slab16 * 4 * L));
  CK(hipMalloc(&ref, slab * 4 * 2)); CK(hipMalloc(&ref16, slab16 * 4 * 2));
  CK(hipMalloc(&W, (size_t)N * N * 4)); CK(hipMalloc(&Wt, (size_t)N * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&ctlw, 64));
  std::vector<float> h(slab), hw((size_t)N * N), hb(N);
  // The chain must stay finite through thousands of relu layers WHATEVER the draw: uniform weights of scale g / sqrt(N) carry a
  // variance gain of g^2 / 6 per layer.  g = 2.45 (gain 1.0004) was a coin toss per draw — and the draw depended on B (one rand()
  // stream for x, W, b), so B = 8, N = 192 diverged to inf - inf = NaN in the probe AND in its launch-per-link reference, and
  // NaN != NaN was counted as 1536 mismatches (round-2 log).  g = 2.3 (gain 0.88): the chain settles at the bias-driven fixed point.
  srand(1);
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f * 2.3f / sqrtf((float)N);
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-4f;
  srand(2 + B);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  CK(hipMemcpy(x0, h.data(), slab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), (size_t)N * N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(x0_16, 0, slab16 * 4));
  hipLaunchKernelGGL(k_t16_pack, dim3((N * N + 255) / 256), dim3(256), 0, 0, W, N, N, Wt);
  hipLaunchKernelGGL(k_t16_pack, dim3((B * N + 255) / 256), dim3(256), 0, 0, x0, B, N, x0_16);
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemsetAsync(slabs16, 0xFF, slab16 * 4 * L, s));
    CK(hipMemsetAsync(ctlw, 0, 64, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((k_persistent<NW>), dim3(G), dim3(NW * 64), 0, s, x0_16, slabs, slabs16, Wt, bias, B, N, L, shift, ctlw);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned hc[16];
  CK(hipMemcpy(hc, ctlw, 64, hipMemcpyDeviceToHost));
  const dim3 grid(N / 16, (B + 15) / 16);  // reference: one launch per link, ping-pong
  CK(hipEventRecord(e0, s));
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL((k_link<NW>), grid, dim3(NW * 64), 0, s, l == 0 ? x0_16 : ref16 + ((l - 1) & 1) * slab16, ref + (l & 1) * slab, ref16 + (l & 1) * slab16, Wt, bias, B, N);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float msl = 0; CK(hipEventElapsedTime(&msl, e0, e1));
  std::vector<float> a(slab), b(slab);
  CK(hipMemcpy(a.data(), slabs + (size_t)(L - 1) * slab, slab * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), ref + ((L - 1) & 1) * slab, slab * 4, hipMemcpyDeviceToHost));
  size_t bad = 0; double mag = 0;
  for (size_t i = 0; i < slab; ++i) { bad += a[i] != b[i]; mag += fabs(b[i]); }
  if (shift == -1) bad = 0;  // compute-only run: results are not the chain's
  if (!(mag == mag) || mag > 1e30) { printf("probe error: the reference chain is not finite (mean|x| %g) — nothing was compared\n", mag / slab); return 1; }
  printf("B=%3d N=K=%4d L=%4d G=%3d NW=%2d shift=%2d: persistent %6.3f us/link | launches %6.3f us/link | mismatches %zu/%zu (mean|x| %.3g) aborted launches=%u code=%u\n",
         B, N, L, G, NW, shift, best * 1e3 / L, msl * 1e3 / L, bad, slab, mag / slab, hc[4], hc[5]);
  (void)hipFree(x0); (void)hipFree(x0_16); (void)hipFree(slabs); (void)hipFree(slabs16); (void)hipFree(ref); (void)hipFree(ref16); (void)hipFree(W); (void)hipFree(Wt);
  (void)hipFree(bias); (void)hipFree(ctlw);
  return 0;
}

// the XCD every workgroup of a 256-workgroup launch runs on (HW_REG_XCC_ID): the placement `shift = -2` relies on
__global__ void k_xcc(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);  // id 20 = XCC_ID, bits [3:0]
}
static int check_xcc() {
  int* d; int h[256];
  CK(hipMalloc(&d, sizeof(h)));
  hipLaunchKernelGGL(k_xcc, dim3(256), dim3(1024), 0, 0, d);
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  int off = 0;
  for (int w = 0; w < 256; ++w) off += h[w] != (w & 7);
  printf("XCC_ID of workgroup w == w %% 8 for %d of 256 workgroups (first eight: %d %d %d %d %d %d %d %d)\n", 256 - off, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  (void)hipFree(d);
  return 0;
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 2000;
  if (argc > 2 && atoi(argv[2]) == 1) {  // the XCD-local hand-off experiment (build with -DPCHAIN_POLL_AUX=... -DPCHAIN_STORE_AUX=...)
    if (check_xcc()) return 1;
    for (int N : {256, 512}) {
      if (run<8>(64, N, L, 256, 0, 3)) return 1;
      if (run<8>(64, N, L, 256, -2, 3)) return 1;
      if (run<16>(64, N, L, 256, 0, 3)) return 1;
      if (run<16>(64, N, L, 256, -2, 3)) return 1;
    }
    return 0;
  }
  for (int B : {8, 64})
    for (int N : {256, 512, 192}) {
      if (run<8>(B, N, L, 256, 0, 3)) return 1;
      if (run<16>(B, N, L, 256, 0, 3)) return 1;
      if (run<8>(B, N, L, 256, 5, 3)) return 1;
      if (run<8>(B, N, L, 256, -1, 3)) return 1;
    }
  return 0;
}
