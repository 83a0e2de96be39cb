// launch_gaps.hip — diagnostic: are there periodic stalls in a long chain of dependent launches WITHOUT a profiler attached?
// Every kernel stamps wall_clock64() at its start and end; the host stamps the return of every launch call.
// Build: hipcc --offload-arch=gfx950 -O3 tools/launch_gaps.hip -o tools/launch_gaps ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <vector>

struct Args { unsigned long long* stamps; int idx; int spin; float pad[52]; };  // 232 bytes like LinArgs<1>

__global__ __launch_bounds__(256) void k_link(Args a) {
  unsigned long long t0 = wall_clock64();
  // ~2.5 us of dependent work, like a chain link
  float x = (float)threadIdx.x;
  for (int i = 0; i < a.spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  if (x == 12345.678f) a.stamps[0] = 0;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[2 * a.idx] = t0; a.stamps[2 * a.idx + 1] = wall_clock64(); }
}

int main(int argc, char** argv) {
  const int N = 20000;
  const int spin = argc > 1 ? atoi(argv[1]) : 600;
  unsigned long long* st;
  if (hipMalloc(&st, sizeof(unsigned long long) * 2 * N) != hipSuccess) return 1;
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return 1;
  std::vector<double> host(N);
  Args a{}; a.stamps = st; a.spin = spin;
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      a.idx = i;
      hipLaunchKernelGGL(k_link, dim3(64), dim3(256), 0, s, a);
      host[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    (void)hipStreamSynchronize(s);
  }
  std::vector<unsigned long long> h(2 * N);
  (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost);
  const double tick_us = 0.01;  // wall_clock64: 100 MHz
  double span = (h[2 * N - 1] - h[0]) * tick_us;
  printf("spin %d: %d launches, GPU span %.1f us = %.3f us/link; host enqueue %.3f us/launch\n", spin, N, span, span / N, host[N - 1] / N);
  int nbig = 0; double sum = 0; int last = -1;
  for (int i = 1; i < N; ++i) {
    double gap = ((double)h[2 * i] - (double)h[2 * i - 1]) * tick_us;
    if (gap > 15.0) { if (nbig < 40) printf("  GPU gap %.1f us before launch %d (+%d)\n", gap, i, last < 0 ? 0 : i - last); last = i; ++nbig; sum += gap; }
  }
  printf("GPU gaps > 15 us: %d, %.1f us in total (%.2f %% of the span)\n", nbig, sum, 100.0 * sum / span);
  nbig = 0; last = -1; sum = 0;
  for (int i = 1; i < N; ++i) {
    double d = host[i] - host[i - 1];
    if (d > 15.0) { if (nbig < 40) printf("  host launch call %.1f us at launch %d (+%d)\n", d, i, last < 0 ? 0 : i - last); last = i; ++nbig; sum += d; }
  }
  printf("host launch calls > 15 us: %d, %.1f us in total\n", nbig, sum);
  return 0;
}
