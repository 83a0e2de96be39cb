// core.hip — error reporting, version and device probe of libblvm_hip.
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace blvm {

namespace {
struct ChainEntry {
  int seen = 0;
  bool bad = false;  // capture failed once: never try again for this key
  hipGraphExec_t exec = nullptr;
  unsigned long long last_use = 0;
};
struct ChainCache {
  std::mutex mu;
  std::unordered_map<std::string, ChainEntry> map;
  unsigned long long tick = 0;
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
};
ChainCache& chain_cache() {
  static ChainCache c;
  return c;
}
bool graphs_enabled() {
  static int v = [] {
    const char* e = getenv("BLVM_GRAPHS");
    return e ? atoi(e) : 0;  // measured slower than plain launches on ROCm 7.2 (see common.h): an experiment switch
  }();
  return v != 0;
}
struct ChainStats {
  unsigned long long plain = 0, captured = 0, replayed = 0, failed = 0;
  ~ChainStats() {
    if (getenv("BLVM_GRAPHS_DEBUG"))
      fprintf(stderr, "[blvm] chain graphs: %llu plain calls, %llu captures, %llu replays, %llu failed captures\n", plain, captured, replayed, failed);
  }
};
ChainStats g_stats;
constexpr int kMaxGraphs = 16;
constexpr size_t kMaxKeys = 1024;
}  // namespace

int run_chain(const ChainKey& key, hipStream_t user, const std::function<int(hipStream_t)>& body) {
  if (!graphs_enabled()) return body(user);
  ChainCache& c = chain_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return body(user);
  if (c.device != dev) {  // one cache per process and device (one process per GPU); a device switch starts over
    for (auto& kv : c.map)
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    c.map.clear();
    if (c.stream) { (void)hipStreamDestroy(c.stream); (void)hipEventDestroy(c.ev_in); (void)hipEventDestroy(c.ev_out); }
    c.stream = nullptr;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c.ev_out, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      c.device = -1;
      return body(user);
    }
    c.device = dev;
  }
  if (c.map.size() > kMaxKeys) {  // forget keys that never repeated
    for (auto it = c.map.begin(); it != c.map.end();) it = it->second.exec ? std::next(it) : c.map.erase(it);
  }
  ChainEntry& e = c.map[key.bytes];
  e.seen++;
  e.last_use = ++c.tick;
  if (e.bad || e.seen < 2) {
    g_stats.plain++;
    return body(user);
  }
  if (e.exec == nullptr) {
    if (hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      e.bad = true;
      return body(user);
    }
    const int rc = body(c.stream);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(c.stream, &g);
    if (rc != BLVM_OK) {  // an argument error found by the body: report it, nothing was enqueued
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      return rc;
    }
    hipGraphExec_t ex = nullptr;
    if (ce != hipSuccess || g == nullptr || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      e.bad = true;
      g_stats.failed++;
      return body(user);
    }
    (void)hipGraphDestroy(g);
    e.exec = ex;
    g_stats.captured++;
    int n = 0;
    for (auto& kv : c.map) n += kv.second.exec != nullptr;
    while (n > kMaxGraphs) {  // drop the least recently used graph
      ChainEntry* old = nullptr;
      for (auto& kv : c.map)
        if (kv.second.exec && &kv.second != &e && (!old || kv.second.last_use < old->last_use)) old = &kv.second;
      if (!old) break;
      (void)hipGraphExecDestroy(old->exec);
      old->exec = nullptr;
      old->seen = 0;
      --n;
    }
  }
  g_stats.replayed++;
  BLVM_HIP(hipEventRecord(c.ev_in, user));
  BLVM_HIP(hipStreamWaitEvent(c.stream, c.ev_in, 0));
  BLVM_HIP(hipGraphLaunch(e.exec, c.stream));
  BLVM_HIP(hipEventRecord(c.ev_out, c.stream));
  BLVM_HIP(hipStreamWaitEvent(user, c.ev_out, 0));
  return BLVM_OK;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace blvm

namespace blvm {
namespace {
struct I32Chunk { int32_t v[256]; };
__global__ __launch_bounds__(256) void upload_i32_kernel(I32Chunk c, int32_t* dst, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = c.v[threadIdx.x];
}
}  // namespace
}  // namespace blvm

// Host integers -> device, carried in kernel arguments (256 per launch): nothing is staged, nothing waits.  A pageable
// host-to-device copy of the batch's lengths blocks the host until the stream has drained, i.e. once per training step it stops
// the host from running ahead of the GPU.
extern "C" int blvm_upload_i32(const int32_t* host, int n, int32_t* dst, void* stream) {
  using namespace blvm;
  BLVM_REQUIRE(n >= 0 && (n == 0 || (host != nullptr && dst != nullptr)), "upload_i32: null pointer");
  for (int off = 0; off < n; off += 256) {
    I32Chunk c;
    const int m = n - off < 256 ? n - off : 256;
    memcpy(c.v, host + off, sizeof(int32_t) * m);
    hipLaunchKernelGGL(upload_i32_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), c, dst + off, m);
  }
  BLVM_CHECK_LAUNCH("upload_i32");
  return BLVM_OK;
}

extern "C" int blvm_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* blvm_last_error(void) { return blvm::g_err; }

extern "C" int blvm_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
