// libgmhip runtime: devices, one compute stream per device, caching allocator,
// copies, events, error reporting.  Plain HIP for gfx950; no torch, no shims.
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <cstdlib>

#include <pthread.h>
#include <unistd.h>

#include "gm_common.h"

namespace gm {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s failed: %s (%d)", what, hipGetErrorString(e), (int)e);
  (void)hipGetLastError();  // clear sticky error
  return (int)e;
}

struct DeviceState {
  std::mutex mu;
  hipStream_t stream = nullptr;
  int cu_count = 0;
  // free blocks by rounded size; live blocks by pointer
  std::multimap<size_t, void*> cached;
  std::unordered_map<void*, size_t> live;
  // staggered blocks: what gm_malloc handed out -> what hipMalloc returned (only where the two differ)
  std::unordered_map<void*, void*> base_of;
  unsigned stagger_turn = 0;
  size_t in_use = 0, cached_bytes = 0;
  bool ready = false;
  // GM_DEBUG_GUARD: what gm_malloc handed out -> (hipMalloc's block, bytes asked for, serial number)
  struct Guarded {
    void* base;
    size_t bytes;
    unsigned long long serial;
  };
  std::unordered_map<void*, Guarded> guarded;
};

// GM_DEBUG_GUARD=1 (debug canaries around device buffers).  Every block of gm_malloc is a hipMalloc of its
// own -- no recycling -- with a poisoned red zone of kGuardZone bytes in front of it and one behind its LAST REQUESTED BYTE
// (the pool's rounding is not the caller's to write into), and it comes filled with 0xFF bytes (float NaN / integer -1: a
// kernel that reads what nobody wrote says so loudly instead of finding the previous tenant's tile).  The zones are checked
// when the block is freed, at every gm_sync and by gm_guard_check; a violation is reported on stderr with the block's size,
// serial number and the first overwritten byte, counted (gm_guard_check), and fails that gm_free / gm_sync with
// GM_ERR_GUARD.  Freed blocks go straight back to the driver (hipFree): a kernel that still uses one faults.  Reads past the
// end of a block are not caught (there are no guard pages); several kernels do read the cells behind a ragged row on purpose.
constexpr size_t kGuardZone = 4096;
constexpr unsigned char kGuardByte = 0xA5;
static bool guard_mode() {
  static const bool on = []() {
    const char* e = getenv("GM_DEBUG_GUARD");
    return e && atoi(e) != 0;
  }();
  return on;
}
static std::atomic<unsigned long long> g_guard_violations{0};
static std::atomic<unsigned long long> g_guard_serial{0};

// zones of one block against the poison byte (stream drained by the caller) -> number of violated zones
static int guard_check_block(void* user, const DeviceState::Guarded& g) {
  static thread_local std::vector<unsigned char> host;
  const size_t tail = ((g.bytes + 255) / 256 * 256 - g.bytes) + kGuardZone;  // (up to the end of hipMalloc's block)
  host.resize(kGuardZone + tail);
  if (hipMemcpy(host.data(), g.base, kGuardZone, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(host.data() + kGuardZone, (char*)user + g.bytes, tail, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int bad = 0;
  for (int zone = 0; zone < 2; ++zone) {
    const unsigned char* z = host.data() + (zone ? kGuardZone : 0);
    const size_t n = zone ? tail : kGuardZone;
    size_t first = n, last = 0, count = 0;
    for (size_t i = 0; i < n; ++i)
      if (z[i] != kGuardByte) {
        if (first == n) first = i;
        last = i;
        ++count;
      }
    if (!count) continue;
    ++bad;
    g_guard_violations.fetch_add(1);
    if (zone)
      fprintf(stderr, "libgmhip GM_DEBUG_GUARD: block #%llu of %zu bytes at %p was written PAST ITS END: %zu bytes differ, the first %zu "
              "and the last %zu bytes behind the block\n", g.serial, g.bytes, user, count, first, last);
    else
      fprintf(stderr, "libgmhip GM_DEBUG_GUARD: block #%llu of %zu bytes at %p was written BEFORE ITS START: %zu bytes differ, the "
              "nearest %zu bytes in front of the block\n", g.serial, g.bytes, user, count, kGuardZone - last);
  }
  return bad;
}

// Tiles of one size come out of hipMalloc with identical low address bits: cell i of A, of B and of the result then
// sit at the same place of the channel / bank interleave, and a 2-read + 1-write stream keeps three rows of the same
// banks open in turn -- and which rows depends on where hipMalloc happened to put the blocks in this process.  Large
// blocks are therefore handed out at a rotating offset of 0 .. 15 steps of 8 KiB inside a block that is 120 KiB larger.
// Measured (profiles/r04_ab_stagger.log, same box, three runs each): the 10-Block mosaic program 370 k (365-377 k) ->
// 378 k Mpix/s (377-379 k), the tiled chain 501 k (482-515 k) -> 500 k (498-502 k), the headline chain 528.8 k ->
// 530.9 k: what it buys is mostly that the run-to-run spread of the address lottery goes away.  Steps of 256 B ..
// 4 KiB were no better or worse (a tiled chain over 64 KiB row pitches lost 2-4 % at 512 B and 4 KiB).
// GM_ALLOC_STAGGER = bytes per step (0 switches it off; a multiple of 256).
constexpr size_t kStaggerFrom = 16u << 20;  // (the runtime hands out 2 MiB granules: below 16 MiB the extra granule would cost > 12 %)
constexpr unsigned kStaggerTurns = 16;
static size_t stagger_step() {
  static const size_t step = []() -> size_t {
    const char* e = getenv("GM_ALLOC_STAGGER");
    if (!e) return 8192;
    const long long v = atoll(e);
    return v > 0 && v % 256 == 0 && v <= (1 << 20) ? (size_t)v : 0;
  }();
  return step;
}
static void free_block(DeviceState& d, void* ptr) {
  auto it = d.base_of.find(ptr);
  if (it != d.base_of.end()) {
    (void)hipFree(it->second);
    d.base_of.erase(it);
  } else {
    (void)hipFree(ptr);
  }
}

static constexpr int kMaxDevices = 64;
static DeviceState g_dev[kMaxDevices];
static std::mutex g_init_mu;
static int g_ndev = -1;

// fork(): the child inherits this library's tables (streams, pools, the device count) but no usable HIP runtime -- its
// first kernel launch or copy would hang or crash somewhere inside the driver.  A child forked after the parent probed
// for devices is therefore marked, and every runtime entry point answers GM_ERR_FORKED in it.  (dask's `processes`
// scheduler forks its workers by default: core/graphs.py:42-49 hands the graph to whatever get_scheduler() returns.)
static std::atomic<bool> g_forked{false};
static int g_parent_pid = 0;
static void mark_forked_child() {
  if (g_ndev >= 0) {
    g_parent_pid = (int)getppid();
    g_forked.store(true, std::memory_order_release);
  }
}
struct ForkMark {
  ForkMark() { pthread_atfork(nullptr, nullptr, mark_forked_child); }
} g_fork_mark;
static inline bool forked() { return g_forked.load(std::memory_order_acquire); }
static int forked_fail() {
  return fail(GM_ERR_FORKED, "forked after gm_init: process %d is a fork() of %d, which had entered libgmhip; a HIP runtime does "
              "not survive fork() -- start workers with spawn (dask_geomodeling_amd.shard starts one fresh process per GPU)",
              (int)getpid(), g_parent_pid);
}

static int probe_devices() {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (g_ndev >= 0) return g_ndev;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (n > kMaxDevices) n = kMaxDevices;
  g_ndev = n;
  return n;
}

static int usable() {
  if (forked()) return forked_fail();
  if (probe_devices() <= 0) return fail(GM_ERR_NO_DEVICE, "no HIP device visible");
  return GM_OK;
}

static int ensure_device(int dev, DeviceState** out) {
  if (int rc = usable()) return rc;
  if (dev < 0 || dev >= g_ndev) return fail(GM_ERR_INVALID, "device %d out of range (0..%d)", dev, g_ndev - 1);
  DeviceState& d = g_dev[dev];
  std::lock_guard<std::mutex> lk(d.mu);
  if (!d.ready) {
    GM_HIP(hipSetDevice(dev));
    GM_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    GM_HIP(hipGetDeviceProperties(&prop, dev));
    d.cu_count = prop.multiProcessorCount;
    d.ready = true;
  }
  *out = &d;
  return GM_OK;
}

// The device chosen with gm_set_device is the PROCESS's device (one worker process per GPU): HIP's own
// current device is per thread and starts at 0, so the pool threads of a threaded dask scheduler would
// otherwise allocate and launch on device 0 while the tiles live on device k.  Every entry point goes
// through current(), which points a thread that has not been pointed yet at the process's device.
static std::atomic<int> g_process_dev{-1};
static thread_local int t_bound_dev = -1;

static int current(DeviceState** out) {
  if (int rc = usable()) return rc;
  int want = g_process_dev.load(std::memory_order_acquire);
  if (want >= 0 && t_bound_dev != want) {
    GM_HIP(hipSetDevice(want));
    t_bound_dev = want;
  }
  int dev = 0;
  GM_HIP(hipGetDevice(&dev));
  return ensure_device(dev, out);
}

hipStream_t current_stream() {
  DeviceState* d = nullptr;
  if (current(&d) != GM_OK) return nullptr;
  return d->stream;
}

int current_cu_count() {
  DeviceState* d = nullptr;
  if (current(&d) != GM_OK) return 0;
  return d->cu_count;
}

static std::atomic<unsigned long long> g_routes[GM_ROUTE_FAMILIES][32];
void route_hit(int family, int slot, unsigned long long n) {
  if (family >= 0 && family < GM_ROUTE_FAMILIES && slot >= 0 && slot < 32) g_routes[family][slot].fetch_add(n, std::memory_order_relaxed);
}
static const int kRouteSlots[GM_ROUTE_FAMILIES] = {GM_RZ_SLOTS, GM_RB_SLOTS, GM_RT_SLOTS, GM_RM_SLOTS, GM_RS_SLOTS};
static const char* const kRouteNames[GM_ROUTE_FAMILIES][32] = {
    {"calls", "direct", "chunk_attempts", "chunk_answered", "chunk_gave_up", "exact_all", "exact_partial", "sort_partition", "sort_radix",
     "select", "lexsort", "block_pipeline", "ordered_sums", "lds_tables", "segment_sort", "lds_exact", "lds_lazy", "exact_lazy_labels", "parallel_lazy", "raw_select"},
    {"calls", "two_x", "staged", "direct", "unstaged_fallback"},
    {"calls", "lookup32_w4", "lookup32_other", "lut8", "lut8_rows", "classify_registers", "classify_lds", "classify_window", "reclassify"},
    {"calls", "network", "lds_sort", "select"},
    {"calls", "hillshade_pairs", "hillshade_single", "movingmax_3x3", "movingmax_stack", "movingmax_columns", "movingmax_keys",
     "dilate8_wide", "dilate_other", "gauss_one_launch", "gauss_two_pass"}};

static size_t round_size(size_t bytes) {
  if (bytes == 0) bytes = 1;
  if (bytes <= (1u << 20)) {  // next power of two, >= 512 B
    size_t r = 512;
    while (r < bytes) r <<= 1;
    return r;
  }
  const size_t g = 2u << 20;  // 2 MiB granules for large tiles
  return (bytes + g - 1) / g * g;
}

}  // namespace gm

using namespace gm;

extern "C" {

int gm_init(void) {
  if (int rc = usable()) return rc;
  DeviceState* d = nullptr;
  return current(&d);
}

int gm_device_count(void) {
  if (forked()) {
    (void)forked_fail();  // (never fails: 0 devices, and gm_last_error() says why)
    return 0;
  }
  return probe_devices();
}

int gm_set_device(int device) {
  if (int rc = usable()) return rc;
  if (device < 0 || device >= g_ndev) return fail(GM_ERR_INVALID, "device %d out of range", device);
  GM_HIP(hipSetDevice(device));
  g_process_dev.store(device, std::memory_order_release);
  t_bound_dev = device;
  DeviceState* d = nullptr;
  return ensure_device(device, &d);
}

int gm_get_device(int* device) {
  if (!device) return fail(GM_ERR_INVALID, "null pointer");
  if (int rc = usable()) return rc;
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  GM_HIP(hipGetDevice(device));
  return GM_OK;
}

int gm_device_pci_bus_id(int device, char* buf, size_t len) {
  if (!buf || len < 13) return fail(GM_ERR_INVALID, "a PCI bus id needs 13 bytes");
  if (int rc = usable()) return rc;
  if (device < 0 || device >= g_ndev) return fail(GM_ERR_INVALID, "device %d out of range", device);
  GM_HIP(hipDeviceGetPCIBusId(buf, (int)len, device));  // (no context on `device` is made for this)
  return GM_OK;
}

int gm_device_name(char* buf, size_t len) {
  if (!buf || len == 0) return fail(GM_ERR_INVALID, "null buffer");
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  int dev = 0;
  GM_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  GM_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, len, "%s %s (%d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return GM_OK;
}

int gm_malloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(GM_ERR_INVALID, "null pointer");
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  size_t sz = round_size(bytes);
  std::lock_guard<std::mutex> lk(d->mu);
  if (guard_mode()) {
    if (bytes == 0) bytes = 1;
    const size_t body = (bytes + 255) / 256 * 256;
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, kGuardZone + body + kGuardZone);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (GM_DEBUG_GUARD)");
    void* user = (char*)base + kGuardZone;
    GM_HIP(hipMemsetAsync(base, kGuardByte, kGuardZone, d->stream));
    GM_HIP(hipMemsetAsync(user, 0xFF, bytes, d->stream));
    GM_HIP(hipMemsetAsync((char*)user + bytes, kGuardByte, body - bytes + kGuardZone, d->stream));
    d->guarded[user] = DeviceState::Guarded{base, bytes, g_guard_serial.fetch_add(1) + 1};
    d->live[user] = sz;
    d->in_use += sz;
    *ptr = user;
    return GM_OK;
  }
  auto it = d->cached.find(sz);
  if (it != d->cached.end()) {
    *ptr = it->second;
    d->cached.erase(it);
    d->cached_bytes -= sz;
  } else {
    const size_t step = sz >= kStaggerFrom ? stagger_step() : 0;
    const size_t extra = step * (kStaggerTurns - 1);
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, sz + extra);
    if (e != hipSuccess) {
      // out of memory: drop the cache and retry once
      (void)hipGetLastError();
      (void)hipStreamSynchronize(d->stream);
      for (auto& kv : d->cached) free_block(*d, kv.second);
      d->cached.clear();
      d->cached_bytes = 0;
      e = hipMalloc(&base, sz + extra);
      if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    }
    *ptr = base;
    if (step) {
      *ptr = (char*)base + step * (d->stagger_turn++ % kStaggerTurns);
      if (*ptr != base) d->base_of[*ptr] = base;
    }
  }
  d->live[*ptr] = sz;
  d->in_use += sz;
  return GM_OK;
}

int gm_malloc_many(void** ptrs, int n, size_t bytes) {
  if (!ptrs || n < 0) return fail(GM_ERR_INVALID, "gm_malloc_many: null table or negative count");
  for (int i = 0; i < n; ++i) {
    const int rc = gm_malloc(&ptrs[i], bytes);
    if (rc != GM_OK) {  // (all or nothing)
      for (int j = 0; j < i; ++j) (void)gm_free(ptrs[j]);
      for (int j = 0; j < n; ++j) ptrs[j] = nullptr;
      return rc;
    }
  }
  return GM_OK;
}

int gm_free(void* ptr) {
  if (!ptr) return GM_OK;
  // the block may belong to any device: search the owner (few devices, cheap)
  for (int i = 0; i < (g_ndev > 0 ? g_ndev : 0); ++i) {
    DeviceState& d = g_dev[i];
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.live.find(ptr);
    if (it == d.live.end()) continue;
    size_t sz = it->second;
    d.live.erase(it);
    d.in_use -= sz;
    auto g = d.guarded.find(ptr);
    if (g != d.guarded.end()) {  // GM_DEBUG_GUARD: checked, then back to the driver (never recycled)
      int bad = 0;
      if (!forked()) {
        (void)hipStreamSynchronize(d.stream);
        bad = guard_check_block(ptr, g->second);
        (void)hipFree(g->second.base);
      }
      const DeviceState::Guarded rec = g->second;
      d.guarded.erase(g);
      if (bad) return fail(GM_ERR_GUARD, "GM_DEBUG_GUARD: block #%llu of %zu bytes was written outside its bounds (see stderr)",
                           rec.serial, rec.bytes);
      return GM_OK;
    }
    d.cached.emplace(sz, ptr);
    d.cached_bytes += sz;
    return GM_OK;
  }
  return fail(GM_ERR_INVALID, "gm_free: pointer %p not allocated by gm_malloc", ptr);
}

// ---- pinned host buffers for results handed back to the host -----------------------------------
// A device-to-host copy into pageable memory runs at ~3 GB/s here (page faults of a fresh
// destination + the runtime's staging); into pinned memory at PCIe speed.  Pinning is expensive,
// so buffers are recycled by size (bounded cache).
namespace {
std::mutex g_host_mu;
std::multimap<size_t, void*> g_host_cached;
std::unordered_map<void*, size_t> g_host_live;
size_t g_host_cached_bytes = 0;
constexpr size_t kHostCacheLimit = 8ull << 30;
}  // namespace

int gm_host_malloc(void** ptr, size_t bytes) {
  if (!ptr) return fail(GM_ERR_INVALID, "null pointer");
  if (int rc = usable()) return rc;
  const size_t sz = round_size(bytes);
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_host_cached.find(sz);
  if (it != g_host_cached.end()) {
    *ptr = it->second;
    g_host_cached.erase(it);
    g_host_cached_bytes -= sz;
  } else {
    hipError_t e = hipHostMalloc(ptr, sz, hipHostMallocNonCoherent);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (auto& kv : g_host_cached) (void)hipHostFree(kv.second);
      g_host_cached.clear();
      g_host_cached_bytes = 0;
      e = hipHostMalloc(ptr, sz, hipHostMallocNonCoherent);
      if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    }
  }
  g_host_live[*ptr] = sz;
  return GM_OK;
}

int gm_host_free(void* ptr) {
  if (!ptr) return GM_OK;
  std::lock_guard<std::mutex> lk(g_host_mu);
  auto it = g_host_live.find(ptr);
  if (it == g_host_live.end()) return fail(GM_ERR_INVALID, "gm_host_free: pointer %p not allocated by gm_host_malloc", ptr);
  const size_t sz = it->second;
  g_host_live.erase(it);
  if (forked()) return GM_OK;  // (the parent's buffer: nothing of the runtime is touched in a forked child)
  if (g_host_cached_bytes + sz > kHostCacheLimit) {
    (void)hipHostFree(ptr);
  } else {
    g_host_cached.emplace(sz, ptr);
    g_host_cached_bytes += sz;
  }
  return GM_OK;
}

int gm_pool_trim(void) {
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(d->mu);
  GM_HIP(hipStreamSynchronize(d->stream));
  for (auto& kv : d->cached) free_block(*d, kv.second);
  d->cached.clear();
  d->cached_bytes = 0;
  return GM_OK;
}

int gm_mem_info(size_t* free_bytes, size_t* total_bytes) {
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  size_t f = 0, t = 0;
  GM_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return GM_OK;
}

int gm_pool_stats(size_t* in_use_bytes, size_t* cached_bytes) {
  DeviceState* d = nullptr;
  int rc = current(&d);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(d->mu);
  if (in_use_bytes) *in_use_bytes = d->in_use;
  if (cached_bytes) *cached_bytes = d->cached_bytes;
  return GM_OK;
}

// Small copies between pageable host memory and the device go through a pinned bounce buffer of the calling thread: the
// runtime's own pageable path costs 30-45 us per copy (a verdict of 16 bytes, the statistics of a thousand labels), the pinned
// one ~10.  Larger copies keep the direct call (callers that care hand over pinned memory: gm_host_malloc / gm_host_register).
namespace {
constexpr size_t kBounceBytes = 256u << 10;
void* bounce_buffer() {
  // (one per thread that copies, never freed: a thread's exit may come after the runtime's own; portable: a thread that
  //  moves to another device with gm_set_device keeps copying through it.  The library is built for one device per process.)
  static thread_local void* buffer = nullptr;
  if (!buffer && hipHostMalloc(&buffer, kBounceBytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    buffer = nullptr;
  }
  return buffer;
}
}  // namespace

int gm_h2d(void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return GM_OK;
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  void* bounce = bytes <= kBounceBytes ? bounce_buffer() : nullptr;
  if (bounce) {
    memcpy(bounce, src_host, bytes);
    GM_HIP(hipMemcpyAsync(dst_dev, bounce, bytes, hipMemcpyHostToDevice, s));
  } else {
    GM_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
  }
  GM_HIP(hipStreamSynchronize(s));
  return GM_OK;
}

int gm_d2h(void* dst_host, const void* src_dev, size_t bytes) {
  if (bytes == 0) return GM_OK;
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  void* bounce = bytes <= kBounceBytes ? bounce_buffer() : nullptr;
  GM_HIP(hipMemcpyAsync(bounce ? bounce : dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
  GM_HIP(hipStreamSynchronize(s));
  if (bounce) memcpy(dst_host, bounce, bytes);
  return GM_OK;
}

int gm_d2h_2d(void* dst_host, size_t dst_pitch_bytes, const void* src_dev, size_t src_pitch_bytes, size_t width_bytes,
              size_t rows) {
  if (!dst_host || !src_dev) return fail(GM_ERR_INVALID, "null pointer");
  if (width_bytes == 0 || rows == 0) return GM_OK;
  if (width_bytes > dst_pitch_bytes || width_bytes > src_pitch_bytes)
    return fail(GM_ERR_INVALID, "gm_d2h_2d: rows of %zu bytes in pitches of %zu / %zu", width_bytes, dst_pitch_bytes, src_pitch_bytes);
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  if (width_bytes == dst_pitch_bytes && width_bytes == src_pitch_bytes)
    GM_HIP(hipMemcpyAsync(dst_host, src_dev, width_bytes * rows, hipMemcpyDeviceToHost, s));
  else
    GM_HIP(hipMemcpy2DAsync(dst_host, dst_pitch_bytes, src_dev, src_pitch_bytes, width_bytes, rows, hipMemcpyDeviceToHost, s));
  return GM_OK;
}

int gm_host_register(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(GM_ERR_INVALID, "gm_host_register: empty range");
  DeviceState* d = nullptr;
  if (int rc = current(&d)) return rc;
  GM_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return GM_OK;
}

int gm_host_unregister(void* ptr) {
  if (!ptr || forked()) return GM_OK;
  DeviceState* d = nullptr;
  if (int rc = current(&d)) return rc;
  GM_HIP(hipStreamSynchronize(d->stream));  // (copies into the range may still be in flight)
  GM_HIP(hipHostUnregister(ptr));
  return GM_OK;
}

int gm_d2d(void* dst_dev, const void* src_dev, size_t bytes) {
  if (bytes == 0) return GM_OK;
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  GM_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, s));
  return GM_OK;
}

int gm_memset(void* dst_dev, int byte_value, size_t bytes) {
  if (bytes == 0) return GM_OK;
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  GM_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, s));
  return GM_OK;
}

int gm_sync(void) {
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  GM_HIP(hipStreamSynchronize(s));
  if (guard_mode()) {
    unsigned long long total = 0;
    return gm_guard_check(&total);
  }
  return GM_OK;
}

int gm_guard_check(unsigned long long* violations_so_far) {
  if (violations_so_far) *violations_so_far = g_guard_violations.load();
  if (!guard_mode()) return GM_OK;
  DeviceState* d = nullptr;
  if (int rc = current(&d)) return rc;
  int bad = 0;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    GM_HIP(hipStreamSynchronize(d->stream));
    for (auto& kv : d->guarded) {
      const int b = guard_check_block(kv.first, kv.second);
      if (b) {  // (reported once: the zones are poisoned again)
        const DeviceState::Guarded& g = kv.second;
        (void)hipMemset(g.base, kGuardByte, kGuardZone);
        (void)hipMemset((char*)kv.first + g.bytes, kGuardByte, (g.bytes + 255) / 256 * 256 - g.bytes + kGuardZone);
      }
      bad += b;
    }
  }
  if (violations_so_far) *violations_so_far = g_guard_violations.load();
  if (bad) return fail(GM_ERR_GUARD, "GM_DEBUG_GUARD: %d red zone(s) of live device blocks were overwritten (see stderr)", bad);
  return GM_OK;
}

int gm_guard_enabled(void) { return guard_mode() ? 1 : 0; }

int gm_event_create(void** event) {
  if (!event) return fail(GM_ERR_INVALID, "null pointer");
  if (!current_stream()) return GM_ERR_NO_DEVICE;
  hipEvent_t ev;
  GM_HIP(hipEventCreate(&ev));
  *event = (void*)ev;
  return GM_OK;
}

int gm_event_destroy(void* event) {
  if (!event || forked()) return GM_OK;
  GM_HIP(hipEventDestroy((hipEvent_t)event));
  return GM_OK;
}

int gm_event_record(void* event) {
  hipStream_t s = current_stream();
  if (!s) return GM_ERR_NO_DEVICE;
  GM_HIP(hipEventRecord((hipEvent_t)event, s));
  return GM_OK;
}

int gm_event_sync(void* event) {
  if (forked()) return forked_fail();
  GM_HIP(hipEventSynchronize((hipEvent_t)event));
  return GM_OK;
}

int gm_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return fail(GM_ERR_INVALID, "null pointer");
  if (forked()) return forked_fail();
  GM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return GM_OK;
}

int gm_route_stats(int family, uint64_t* counters, int n) {
  if (family < 0 || family >= GM_ROUTE_FAMILIES) return -1;
  for (int i = 0; counters && i < n && i < kRouteSlots[family]; ++i) counters[i] = g_routes[family][i].load(std::memory_order_relaxed);
  return kRouteSlots[family];
}

const char* gm_route_name(int family, int slot) {
  if (family < 0 || family >= GM_ROUTE_FAMILIES || slot < 0 || slot >= kRouteSlots[family]) return nullptr;
  return kRouteNames[family][slot];
}

const char* gm_last_error(void) { return g_err.c_str(); }
const char* gm_version(void) { return GM_VERSION; }

}  // extern "C"
