// cu_ingest — how fast can ONE workgroup pull a weight stream out of L2 / Infinity Cache?  (design probe for the
// single-workgroup WaveNet decoder: its per-block time is set by this rate, not by MFMA or LDS.)
// Patterns: 0 = MFMA-operand pattern of the decoder (16 rows x 64 B per wave instruction, rows 512 B apart);
//           1 = fully coalesced (lane-contiguous 16 B: 1 KB per wave instruction).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/cu_ingest tools/cu_ingest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PATTERN, int NLOADS>
__global__ __launch_bounds__(512) void ingest(const float* __restrict__ w, size_t n_chunks, size_t chunk_floats, int iters, float* out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, cc = lane & 15;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    for (size_t ch = 0; ch < n_chunks; ++ch) {
      const float* base = w + ch * chunk_floats;
      float4 v[NLOADS];
#pragma unroll
      for (int j = 0; j < NLOADS; ++j) {
        const float* p;
        if (PATTERN == 0) p = base + (size_t)(wave * 16 + cc) * (NLOADS * 16) + 4 * q + 16 * j;   // [128 rows][NLOADS*16]
        else p = base + (size_t)(j * 8 + wave) * 256 + lane * 4;
        v[j] = *reinterpret_cast<const float4*>(p);
      }
#pragma unroll
      for (int j = 0; j < NLOADS; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

template <int P, int NL>
void run(const float* w, size_t total_floats, int wgs, float* out) {
  const size_t chunk = 128 * NL * 16;  // floats per chunk (NL=12 -> 96 KB)
  const size_t n_chunks = total_floats / chunk;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((ingest<P, NL>), dim3(wgs), dim3(512), 0, 0, w, n_chunks, chunk, 2, out);
  hipDeviceSynchronize();
  const int iters = 200;
  hipEventRecord(a);
  hipLaunchKernelGGL((ingest<P, NL>), dim3(wgs), dim3(512), 0, 0, w, n_chunks, chunk, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)n_chunks * chunk * 4 * iters;
  printf("pattern %d  loads/lane %2d  footprint %5.2f MB  wgs %3d : %7.1f GB/s per WG, %6.2f us per 96 KB\n", P, NL, n_chunks * chunk * 4 / 1048576.0, wgs,
         bytes / ms / 1e6, ms * 1e3 / (n_chunks * (double)iters) * (98304.0 / (chunk * 4)));
}

int main() {
  const size_t maxf = 64u << 20;
  float *w, *out;
  hipMalloc(&w, maxf * 4);
  hipMalloc(&out, 64);
  hipMemset(w, 0, maxf * 4);
  for (size_t mb : {1, 2, 4, 5, 16, 64}) {
    run<0, 12>(w, mb * 262144, 1, out);
    run<1, 12>(w, mb * 262144, 1, out);
  }
  run<0, 12>(w, 5 * 262144, 8, out);
  run<0, 12>(w, 5 * 262144, 64, out);
  run<1, 12>(w, 5 * 262144, 64, out);
  run<1, 12>(w, 5 * 262144, 256, out);
  return 0;
}
