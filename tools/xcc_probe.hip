// Diagnostic: which XCD does block b of consecutive launches land on?  (hipcc --offload-arch=gfx950 -O2 xcc_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.y * gridDim.x + blockIdx.x] = (int)(x & 0xf);
  }
}
int main() {
  int* d;
  hipMalloc(&d, 4096 * sizeof(int));
  std::vector<int> h(4096);
  const int grids[][2] = {{16, 4}, {32, 4}, {96, 4}, {16, 4}, {12, 1}, {16, 4}, {16, 4}, {3, 1}, {16, 4}, {16, 4}};
  for (int rep = 0; rep < 2; ++rep)
    for (auto& g : grids) {
      hipLaunchKernelGGL(probe, dim3(g[0], g[1]), dim3(256), 0, 0, d);
      hipMemcpy(h.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
      int n = g[0] * g[1], rr = 1;
      for (int b = 0; b < n; ++b) rr &= (h[b] == (h[0] + b) % 8);
      printf("grid %3dx%d (%4d blocks): block0 -> XCD %d, blocks 0..9:", g[0], g[1], n, h[0]);
      for (int b = 0; b < 10 && b < n; ++b) printf(" %d", h[b]);
      printf("  strict round-robin: %s\n", rr ? "yes" : "NO");
    }
  // back-to-back without host sync
  for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(probe, dim3(16, 4), dim3(256), 0, 0, d + i * 64);
  hipMemcpy(h.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
  printf("6 async launches of 64 blocks: block0 XCDs:");
  for (int i = 0; i < 6; ++i) printf(" %d", h[i * 64]);
  printf("\n");
  return 0;
}
