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
