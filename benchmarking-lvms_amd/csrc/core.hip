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

// ---- persistent-chain control block (pchain.h) ---------------------------------------------------------------------------------
// One per process and device: 64 bytes of device memory (word 0 = epoch of the last aborted launch) and 64 bytes of pinned host
// memory mapped into the device (word 0 = aborted launches so far, word 1 = code of the last failed spin), written by a kernel
// only when a bounded spin gives up — the normal path costs nothing and the API stays asynchronous.
namespace blvm {
namespace {
struct PchainCtl {
  std::mutex mu;
  int device = -1;
  unsigned* dev = nullptr;
  unsigned* host = nullptr;      // host view
  unsigned* host_dev = nullptr;  // device view of the same words
  unsigned epoch = 0;
  unsigned taken = 0;  // aborted launches already reported through blvm_async_errors_take()
};
PchainCtl& pchain_ctl_state() {
  static PchainCtl c;
  return c;
}
}  // namespace

namespace {
int g_pchain_max_b = -1, g_pchain_nw = -1, g_pchain_tune = -1;
unsigned long long* g_pchain_prof = nullptr;
}
unsigned long long* pchain_profile_buffer() { return g_pchain_prof; }
int pchain_max_batch() {
  if (g_pchain_max_b < 0) {
    const char* e = getenv("BLVM_PCHAIN");
    const char* m = getenv("BLVM_PCHAIN_MAX_B");
    g_pchain_max_b = (e && atoi(e) == 0) ? 0 : (m ? atoi(m) : 128);
    // a CU mask takes CUs away behind the runtime's back: the persistent launches (which need every workgroup resident) are off
    // and the recurrent chains run as one launch per link
    for (const char* v : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"})
      if (const char* cm = getenv(v); cm && *cm) g_pchain_max_b = 0;
  }
  return g_pchain_max_b;
}
static int g_operand_dtype = -1;
bool operand_bf16() {
  if (g_operand_dtype < 0) {
    const char* e = getenv("BLVM_DTYPE");
    g_operand_dtype = (e && (strcmp(e, "bf16") == 0 || strcmp(e, "1") == 0)) ? 1 : 0;
  }
  return g_operand_dtype == 1;
}
int pchain_waves() {
  if (g_pchain_nw < 0) {
    // 16 waves (a tile's K split 16 ways) since round 3: same-box A/B of the train steps, 8 -> 16 waves: VRNN 15.59 -> 15.50 ms,
    // SRNN 15.71 -> 15.59, CW-VAE (K = 192: one chunk per wave) 87.5 -> 86.1.  BLVM_PCHAIN_NW=8 selects the former default.
    const char* e = getenv("BLVM_PCHAIN_NW");
    g_pchain_nw = (e && atoi(e) == 8) ? 8 : 16;
  }
  return g_pchain_nw;
}

int pchain_tune() {
  if (g_pchain_tune < 0) {
    const char* e = getenv("BLVM_PCHAIN_TUNE");
    g_pchain_tune = e ? atoi(e) : 20;
  }
  return g_pchain_tune;
}

int pchain_ctl(unsigned** dev, unsigned** host_dev, unsigned* epoch) {
  PchainCtl& c = pchain_ctl_state();
  std::lock_guard<std::mutex> lock(c.mu);
  int d = 0;
  BLVM_HIP(hipGetDevice(&d));
  if (c.device != d) {  // one process per GPU; a device switch starts over (the old blocks are leaked on purpose: a kernel may still use them)
    BLVM_HIP(hipMalloc(reinterpret_cast<void**>(&c.dev), 64));
    BLVM_HIP(hipMemset(c.dev, 0, 64));
    BLVM_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.host), 64, hipHostMallocMapped));
    memset(c.host, 0, 64);
    BLVM_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c.host_dev), c.host, 0));
    c.device = d;
    c.epoch = 0;
    c.taken = 0;
  }
  *dev = c.dev;
  *host_dev = c.host_dev;
  *epoch = ++c.epoch;
  if (c.epoch == 0) *epoch = ++c.epoch;  // 0 is the block's reset value
  return BLVM_OK;
}
}  // namespace blvm

extern "C" int blvm_pchain_configure(int max_batch, int waves) {
  if (max_batch >= 0) blvm::g_pchain_max_b = max_batch;
  if (waves == 8 || waves == 16) blvm::g_pchain_nw = waves;
  return BLVM_OK;
}

extern "C" int blvm_set_operand_dtype(int dtype) {
  if (dtype != BLVM_DTYPE_F32 && dtype != BLVM_DTYPE_BF16) {
    blvm::set_error("blvm_set_operand_dtype: %d is neither BLVM_DTYPE_F32 nor BLVM_DTYPE_BF16", dtype);
    return BLVM_EINVAL;
  }
  blvm::g_operand_dtype = dtype;
  return BLVM_OK;
}
extern "C" int blvm_get_operand_dtype(void) { return blvm::operand_bf16() ? BLVM_DTYPE_BF16 : BLVM_DTYPE_F32; }

extern "C" int blvm_pchain_max_batch(void) { return std::min(blvm::pchain_max_batch(), 128); }

extern "C" int blvm_pchain_tune(int bits) {
  blvm::g_pchain_tune = bits;
  return BLVM_OK;
}

extern "C" int blvm_pchain_profile(unsigned long long* device_buffer) {
  blvm::g_pchain_prof = device_buffer;
  return BLVM_OK;
}

extern "C" int blvm_async_errors(unsigned* last_code) {
  blvm::PchainCtl& c = blvm::pchain_ctl_state();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.host) return 0;
  const volatile unsigned* h = c.host;
  if (last_code) *last_code = h[1];
  return (int)h[0];
}

// Aborted launches since the previous take (read-and-clear view of the same counter): a caller that has handled an abort — dropped
// the step, re-run it — is not told about it again by every later check.
extern "C" int blvm_async_errors_take(unsigned* last_code) {
  blvm::PchainCtl& c = blvm::pchain_ctl_state();
  std::lock_guard<std::mutex> lock(c.mu);
  if (!c.host) return 0;
  const volatile unsigned* h = c.host;
  const unsigned total = h[0];
  const unsigned fresh = total - c.taken;
  c.taken = total;
  if (last_code) *last_code = fresh ? h[1] : 0u;
  return (int)fresh;
}

extern "C" int blvm_version(void) { return 130; /* 0.1.3: blvm_wgrad_group_f32 */ }

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
