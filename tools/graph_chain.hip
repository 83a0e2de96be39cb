// graph_chain — does a captured hipGraph run a chain of dependent kernels faster than plain stream launches?
// (a recurrent step is ~18 dependent links; each costs one kernel boundary.)  build: hipcc --offload-arch=gfx950 -O3 -o tools/graph_chain tools/graph_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_touch(const float* in, float* out) { if (threadIdx.x == 0) out[blockIdx.x] = in[blockIdx.x] + 1.f; }
__global__ void k_empty() {}
struct Big { const float* A[3]; const float* W[3]; const float* bias[3]; float* out[3]; int lda[3], ldw[3], ldo[3], tiles[3], flags[3], K[3]; int B; float slope; };
__global__ void k_big(Big a) {  // argument block the size of a 3-segment link
  const int s = blockIdx.x >= a.tiles[0] ? 1 : 0;
  const float* in = s ? a.A[1] : a.A[0];
  float* out = s ? a.out[1] : a.out[0];
  if (threadIdx.x == 0) out[blockIdx.x] = in[blockIdx.x] + a.slope;
}

int main() {
  const int N = 2000;
  float *a, *b;
  CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
  CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto chain = [&](hipStream_t st, bool empty) {
    for (int i = 0; i < N; ++i) {
      if (empty) hipLaunchKernelGGL(k_empty, dim3(128), dim3(256), 0, st);
      else hipLaunchKernelGGL(k_touch, dim3(128), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b);
    }
  };
  {
    Big g{}; g.tiles[0] = 64; g.slope = 1.f;
    auto chain_big = [&](hipStream_t st) {
      for (int i = 0; i < N; ++i) {
        g.A[0] = g.A[1] = (i & 1) ? b : a; g.out[0] = g.out[1] = (i & 1) ? a : b;
        hipLaunchKernelGGL(k_big, dim3(128), dim3(256), 0, st, g);
      }
    };
    chain_big(s); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); chain_big(s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("big-arg chain, stream launches : %.3f us/kernel\n", ms * 1e3 / N);
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain_big(s);
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("big-arg chain, one hipGraph    : %.3f us/kernel\n", ms * 1e3 / N);
  }
  for (int empty = 1; empty >= 0; --empty) {
    chain(s, empty); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); chain(s, empty); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s chain, stream launches : %.3f us/kernel\n", empty ? "empty" : "touch", ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    chain(s, empty);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s chain, one hipGraph    : %.3f us/kernel\n", empty ? "empty" : "touch", ms * 1e3 / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
