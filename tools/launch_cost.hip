// launch_cost.hip — diagnostic: how does the cost of one link of a dependent launch chain depend on the kernel's executed
// code size, its VGPR allocation, its LDS allocation and its workgroup size?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

template <int NOPS>  // NOPS x 4 bytes of straight-line code executed by every wave
__global__ void k_code(float* p) {
  asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(NOPS));
  if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}
template <int NOPS>  // the same bytes present in the kernel but jumped over
__global__ void k_code_skipped(float* p, int never) {
  if (never) asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(NOPS));
  if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}
__global__ void k_vgpr128(float* p) { asm volatile("v_mov_b32 v127, 0" ::: "v127"); if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void k_vgpr250(float* p) { asm volatile("v_mov_b32 v250, 0" ::: "v250"); if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
template <int BYTES>
__global__ void k_lds(float* p) {
  __shared__ float s[BYTES / 4];
  s[threadIdx.x] = 1.f;
  __syncthreads();
  if (threadIdx.x == 0) p[blockIdx.x] += s[1];
}

int main() {
  const int REP = 2000;
  float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 50; ++i) launch();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < REP; ++i) launch();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %7.3f us/launch\n", name, ms * 1e3 / REP);
  };
  for (int wg : {64, 128, 384}) {
    char nm[96];
    snprintf(nm, sizeof nm, "touch, %d WG x 256", wg);
    timeit(nm, [&] { hipLaunchKernelGGL(k_touch, dim3(wg), dim3(256), 0, s, buf); });
    snprintf(nm, sizeof nm, "touch, %d WG x 1024", wg);
    timeit(nm, [&] { hipLaunchKernelGGL(k_touch, dim3(wg), dim3(1024), 0, s, buf); });
  }
#define CODE(N) timeit("executed straight-line code " #N " x 4 B, 128 WG x 256", [&] { hipLaunchKernelGGL((k_code<N>), dim3(128), dim3(256), 0, s, buf); });
  CODE(256) CODE(512) CODE(1024) CODE(2048) CODE(4096) CODE(8192)
#define SKIP(N) timeit("same bytes present but skipped  " #N " x 4 B, 128 WG x 256", [&] { hipLaunchKernelGGL((k_code_skipped<N>), dim3(128), dim3(256), 0, s, buf, 0); });
  SKIP(1024) SKIP(4096)
  timeit("executed code 1024 x 4 B, 128 WG x 1024", [&] { hipLaunchKernelGGL((k_code<1024>), dim3(128), dim3(1024), 0, s, buf); });
  timeit("touch + 128 VGPRs, 128 WG x 256", [&] { hipLaunchKernelGGL(k_vgpr128, dim3(128), dim3(256), 0, s, buf); });
  timeit("touch + 251 VGPRs, 128 WG x 256", [&] { hipLaunchKernelGGL(k_vgpr250, dim3(128), dim3(256), 0, s, buf); });
  timeit("touch + 128 VGPRs, 128 WG x 1024", [&] { hipLaunchKernelGGL(k_vgpr128, dim3(128), dim3(1024), 0, s, buf); });
  timeit("touch + 16 KB LDS, 128 WG x 256", [&] { hipLaunchKernelGGL((k_lds<16384>), dim3(128), dim3(256), 0, s, buf); });
  timeit("touch + 64 KB LDS, 128 WG x 256", [&] { hipLaunchKernelGGL((k_lds<65536>), dim3(128), dim3(256), 0, s, buf); });
  return 0;
}
