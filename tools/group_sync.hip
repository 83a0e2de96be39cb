// group_sync.hip — diagnostic: what does a barrier + small data exchange among 16 workgroups that share one XCD cost,
// compared with the 2.7 us launch boundary?  (Would fusing consecutive MLP links of a recurrent step into one launch pay?)
// hipcc --offload-arch=gfx950 -O3 -w group_sync.hip -o group_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// grid = 8 * WGS blocks; block b runs on XCD b % 8 (observed: strict round-robin from XCD 0).  Groups = XCDs 0..G-1, each with
// WGS workgroups (slot = b / 8).  Per "layer": every workgroup writes 64 floats of "activations", arrives on its group's
// counter, waits for all WGS arrivals, then reads all WGS*64 floats written by the group (L1-bypassing loads).
template <int WGS>
__global__ __launch_bounds__(256) void k_group(float* buf, unsigned* counters, int groups, int layers, float* sink) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (xcd >= groups) return;
  float* act = buf + (size_t)xcd * 2 * WGS * 64;  // double-buffered by layer parity
  unsigned* cnt = counters + xcd * 64;            // one cache line per group
  float acc = 0.f;
  for (int l = 0; l < layers; ++l) {
    float* cur = act + (l & 1) * WGS * 64;
    if (threadIdx.x < 64) __hip_atomic_store(cur + slot * 64 + threadIdx.x, acc + (float)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 write-through store
    __builtin_amdgcn_s_waitcnt(0);  // stores issued ... (vmcnt(0))
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // data stores are write-through and drained (vmcnt(0)) above
      const unsigned target = (unsigned)WGS * (l + 1);
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // read the whole group's activations bypassing L1 (device-scope relaxed atomic loads = sc1 loads)
    float s = 0.f;
    for (int i = threadIdx.x; i < WGS * 64; i += 256) s += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += s * 1e-6f;
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main() {
  float *buf, *sink; unsigned* cnt;
  CK(hipMalloc(&buf, 1 << 20)); CK(hipMalloc(&sink, 1 << 16)); CK(hipMalloc(&cnt, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int groups : {1, 4, 8}) {
    for (int layers : {1, 2, 4, 8, 16}) {
      const int REP = 500;
      auto run = [&](int n) {
        for (int i = 0; i < n; ++i) {
          hipMemsetAsync(cnt, 0, 4096, s);
          hipLaunchKernelGGL((k_group<16>), dim3(8 * 16), dim3(256), 0, s, buf, cnt, groups, layers, sink);
        }
      };
      run(20);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      run(REP);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("groups=%d x 16 WGs, %2d fused layers: %7.3f us per launch (incl. the counter memset launch)\n", groups, layers, ms * 1e3 / REP);
    }
  }
  return 0;
}
