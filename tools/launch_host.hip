// launch_host — host-side cost of enqueueing one small kernel, by launch API and argument size (the B = 64 chains are bound by it).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/launch_host tools/launch_host.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { const float* A[3]; const float* W[3]; const float* bias[3]; float* out[3]; int lda[3], ldw[3], ldo[3], tiles[3], flags[3], K[3]; int B; float slope; };
__global__ void k_big(Big a) { if (threadIdx.x == 0 && a.out[0]) a.out[0][blockIdx.x] = a.slope; }
__global__ void k_ptr(float* p, float v) { if (threadIdx.x == 0 && p) p[blockIdx.x] = v; }
__global__ void k_none() {}
// argument-less launches: the argument block lives in a __device__ table, the step index in a __device__ counter that a
// one-wave kernel bumps once per "step" of 9 links
struct Tab { Big e[16]; long long stride[16]; };
__device__ Tab g_tab;
__device__ unsigned g_step;
template <int SLOT>
__global__ void k_tab() {
  const unsigned t = g_step;
  const Big& a = g_tab.e[SLOT];
  float* out = a.out[0] + t * g_tab.stride[SLOT];
  if (threadIdx.x == 0 && out) out[blockIdx.x] = a.slope + t;
}
__global__ void k_bump() { if (threadIdx.x == 0) g_step = g_step + 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 4000;
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  Big g{}; g.out[0] = buf; g.slope = 1.f;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 200; ++i) launch();
    hipStreamSynchronize(s);
    const double t0 = now();
    for (int i = 0; i < N; ++i) launch();
    const double t1 = now();
    hipStreamSynchronize(s);
    const double t2 = now();
    printf("%-52s host %.2f us/launch, total %.2f us/launch\n", name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  };
  run("hipLaunchKernelGGL, no args, 128x256", [&] { hipLaunchKernelGGL(k_none, dim3(128), dim3(256), 0, s); });
  run("hipLaunchKernelGGL, no args, 1x64", [&] { hipLaunchKernelGGL(k_none, dim3(1), dim3(64), 0, s); });
  run("hipLaunchKernelGGL, 12-byte args", [&] { hipLaunchKernelGGL(k_ptr, dim3(128), dim3(256), 0, s, buf, 1.f); });
  run("hipLaunchKernelGGL, 232-byte struct", [&] { hipLaunchKernelGGL(k_big, dim3(128), dim3(256), 0, s, g); });
  {
    void* args[] = {&g};
    run("hipLaunchKernel (void** args), 232-byte struct", [&] { (void)hipLaunchKernel(reinterpret_cast<const void*>(k_big), dim3(128), dim3(256), args, 0, s); });
  }
  {
    hipFunction_t f = nullptr;
    hipError_t e = hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(k_big));
    if (e == hipSuccess && f) {
      size_t sz = sizeof(Big);
      void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &g, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
      run("hipModuleLaunchKernel (packed buffer), 232-byte struct", [&] { (void)hipModuleLaunchKernel(f, 128, 1, 1, 256, 1, 1, 0, s, nullptr, extra); });
    } else printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(e));
  }
  {
    Tab h{}; for (int i = 0; i < 16; ++i) { h.e[i] = g; h.stride[i] = 0; }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &h, sizeof h));
    unsigned z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_step), &z, sizeof z));
    int n = 0;
    run("argument-less launch, args from a __device__ table (+bump/9)", [&] {
      hipLaunchKernelGGL(k_tab<3>, dim3(128), dim3(256), 0, s);
      if (++n % 9 == 0) hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, s);
    });
  }
  return 0;
}
