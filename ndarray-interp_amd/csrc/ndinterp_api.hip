// csrc/ndinterp_api.hip -- the extern "C" surface of libndinterp_hip.so (include/ndinterp.h).
//
// Host side of the drop-in boundary: owns the device copies of knots / data / spline tables,
// validates like the reference's builders, sequences the kernels of kernels.hpp on a HIP stream
// and translates the device status word into the reference's error values.
// There is deliberately NO CPU evaluation path in this library.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "common.hpp"
#include "host_logic.hpp"
#include "kernels.hpp"

#define NDI_API extern "C" __attribute__((visibility("default")))

namespace ndi {

// ---------------------------------------------------------------------------------------------
// small RAII helpers
// ---------------------------------------------------------------------------------------------
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    NDI_HIP(hipGetDevice(&prev));
    if (prev != dev) NDI_HIP(hipSetDevice(dev));
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;   // false: a view into another allocation (adopt): released by forgetting it
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  // A slice of a larger allocation that outlives this buffer (small handles keep all their tables in ONE allocation:
  // every hipMalloc costs as much as the kernels of a small build).  reserve() beyond the slice falls back to an
  // allocation of its own.
  void adopt(void* ptr, size_t nbytes) {
    release();
    p = ptr;
    bytes = nbytes;
    owned = false;
  }
  void reserve(size_t need) {
    if (need <= bytes) return;
    release();
    NDI_HIP(hipMalloc(&p, need));
    bytes = need;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// Device-to-device copy of a replica's tables (ndi_interp{1,2}d_clone).  Between two GPUs that can address each
// other (xGMI peers) it is one hipMemcpyPeer; when hipDeviceCanAccessPeer says no -- or NDI_CLONE_STAGED=1 forces it,
// which is how the 1-GPU test box exercises the path -- the bytes are staged through a pinned host buffer in
// 64 MiB pieces (device -> host on the source device, host -> device on the destination).
static void copy_across_devices(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes) {
  if (bytes == 0) return;
  const char* force = std::getenv("NDI_CLONE_STAGED");   // read per call: a test switches it
  int can = 1;
  if (dst_dev != src_dev) NDI_HIP(hipDeviceCanAccessPeer(&can, dst_dev, src_dev));
  if (can && !(force && force[0] == '1')) {
    NDI_HIP(hipMemcpyPeer(dst, dst_dev, src, src_dev, bytes));
    return;
  }
  constexpr size_t PIECE = 64ull << 20;
  void* pin = nullptr;
  NDI_HIP(hipHostMalloc(&pin, std::min(bytes, PIECE), hipHostMallocPortable));
  struct Free { void* p; ~Free() { (void)hipHostFree(p); } } guard{pin};
  for (size_t off = 0; off < bytes; off += PIECE) {
    const size_t nb = std::min(PIECE, bytes - off);
    {
      DeviceGuard g(src_dev);
      NDI_HIP(hipMemcpy(pin, (const char*)src + off, nb, hipMemcpyDeviceToHost));
    }
    {
      DeviceGuard g(dst_dev);
      NDI_HIP(hipMemcpy((char*)dst + off, pin, nb, hipMemcpyHostToDevice));
    }
  }
}

// Test hook (tests/test_gpu_short_rows.py): NDI_TEST_FAIL_LAZY_ALLOC=1 makes the lazily built optional copies (bucket
// indices, interval-packed tables) fail as an out-of-memory hipMalloc would -- the fallbacks must serve the batch.
static void maybe_fail_lazy_alloc(int line) {
  const char* e = std::getenv("NDI_TEST_FAIL_LAZY_ALLOC");
  if (e && e[0] == '1') throw HipFailure{hipErrorOutOfMemory, "injected lazy-allocation failure", line};
}

// Workspaces are keyed by (stream, calling thread): work enqueued on one stream is ordered, so a
// thread may reuse its scratch across stream-ordered evaluations, and two host threads that share a
// stream (e.g. the default stream) still get private scratch -- `eval` on one handle is re-entrant.
struct SpaceKey {
  hipStream_t stream;
  std::thread::id tid;
  bool operator<(const SpaceKey& o) const {
    return stream != o.stream ? std::less<hipStream_t>()(stream, o.stream) : tid < o.tid;
  }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function on the *current device*: one flag per
// (function, device), and the return code is checked (a process may drive several devices, one handle each).
static void allow_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  NDI_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  if (done.count({fn, dev})) return;
  NDI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({fn, dev});
}

// roctx ranges around build / locate / group / evaluate (rocprofv3 --marker-trace).  The marker library is
// resolved at run time and only when NDI_ROCTX=1, so the product has no link-time dependency on a profiler.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* on = std::getenv("NDI_ROCTX");
    if (!on || on[0] == '0') return;
    void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static const Roctx& roctx() {
  static Roctx r;
  return r;
}
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~Range() {
    if (on) roctx().pop();
  }
};

// ---------------------------------------------------------------------------------------------
// per-kernel event profiling (ndi_profile_*)
// ---------------------------------------------------------------------------------------------
enum ProfCat : int { PC_EVAL = 0, PC_LOCATE = 1, PC_GROUP = 2 };
struct ProfRec { hipEvent_t a, b; int cat; int dev; };
static std::atomic<int> g_prof_on{0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static ndi_profile g_prof_acc{};
static std::atomic<int> g_last_path{0};

static const char* const PROF_NAMES[3] = {"ndi:evaluate", "ndi:locate", "ndi:group"};
// Timing events are recycled (creating two events per launch costs the host ~0.1 ms per ring step, and the host
// path of chunk 0 is exposed): finished records hand their events back to this pool (g_prof_mu held).  Events
// belong to a device: the pool is keyed by it.
static std::map<int, std::vector<hipEvent_t>> g_prof_pool;
static hipEvent_t prof_event_locked(int dev) {
  auto& pool = g_prof_pool[dev];
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  NDI_HIP(hipEventCreate(&e));
  return e;
}
struct ProfScope {
  hipStream_t s;
  ProfRec r{};
  int dev = 0;
  bool on;
  Range range;   // host-side enqueue range of the stage (the kernels inside carry their own names)
  ProfScope(hipStream_t stream, int cat) : s(stream), on(g_prof_on.load() != 0), range(PROF_NAMES[cat]) {
    if (!on) return;
    r.cat = cat;
    NDI_HIP(hipGetDevice(&dev));
    {
      std::lock_guard<std::mutex> g(g_prof_mu);
      r.a = prof_event_locked(dev);
      r.b = prof_event_locked(dev);
    }
    r.dev = dev;
    NDI_HIP(hipEventRecord(r.a, s));
  }
  void done() {
    if (!on) return;
    NDI_HIP(hipEventRecord(r.b, s));
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof_recs.push_back(r);
    if (g_prof_recs.size() >= 4096) fold_finished_locked();   // profiling left on and never read: stay bounded
  }
  static void account_locked(const ProfRec& r, float ms) {
    if (r.cat == PC_EVAL) { g_prof_acc.eval_launches++; g_prof_acc.eval_ms += ms; }
    else if (r.cat == PC_LOCATE) { g_prof_acc.locate_launches++; g_prof_acc.locate_ms += ms; }
    else { g_prof_acc.group_launches++; g_prof_acc.group_ms += ms; }
  }
  // Adds the records whose kernels have finished to the running totals and recycles their events (g_prof_mu held).
  static void fold_finished_locked() {
    size_t keep = 0;
    for (size_t i = 0; i < g_prof_recs.size(); ++i) {
      ProfRec& r = g_prof_recs[i];
      if (hipEventQuery(r.b) != hipSuccess) {
        (void)hipGetLastError();
        g_prof_recs[keep++] = r;
        continue;
      }
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) account_locked(r, ms);
      g_prof_pool[r.dev].push_back(r.a);
      g_prof_pool[r.dev].push_back(r.b);
    }
    g_prof_recs.resize(keep);
  }
};

// ---------------------------------------------------------------------------------------------
// knot pyramid on the device
// ---------------------------------------------------------------------------------------------
template <class T>
struct DevicePyramid {
  DevBuf buf;
  Pyramid<T> view{};
  std::vector<T> host_knots;
  size_t lds_bytes = 0;

  void upload(const T* knots, uint64_t n) {
    host_knots.assign(knots, knots + n);
    uint32_t block = 1;
    while ((uint64_t)64 * block < n) block *= 2;          // top level <= 64 entries: one per lane
    const uint32_t n1 = (uint32_t)((n + block - 1) / block);
    std::vector<T> all(n + n1);
    std::copy(knots, knots + n, all.begin());
    for (uint32_t j = 0; j < n1; ++j) all[n + j] = knots[(uint64_t)j * block];
    buf.reserve(all.size() * sizeof(T));
    NDI_HIP(hipMemcpy(buf.p, all.data(), all.size() * sizeof(T), hipMemcpyHostToDevice));
    view.lv0 = buf.as<T>();
    view.lv1 = view.lv0 + n;
    view.n = (uint32_t)n;
    view.n1 = n1;
    view.block = block;
    view.levels = (n <= 64) ? 1 : 2;
    // evenly spaced within less than half a step everywhere -> the reference's O(1) guess will mostly be right
    bool even = n >= 2;
    const double step = ((double)knots[n - 1] - (double)knots[0]) / (double)(n - 1);
    for (uint64_t i = 0; even && i < n; ++i)
      even = std::fabs((double)knots[i] - ((double)knots[0] + step * (double)i)) < 0.45 * step;
    view.guess = even ? 1 : 0;
    lds_bytes = all.size() * sizeof(T);
    // Is the formula guess right for EVERY x (then no bucket index is needed)?  The guess is monotone in x, so it is
    // enough that it maps every knot k[i] and the last value before k[i+1] to i (same T arithmetic as locate_index).
    bool exact = even;
    if (exact) {
      const T k0 = knots[0], kn = knots[n - 1];
      auto guess = [&](T x) -> uint64_t {
        const T m = (T(n - 1u) - T(0)) / (kn - k0) * (x - k0) + T(0);
        return (m >= T(0)) ? (uint64_t)(m < T(n - 2u) ? m : T(n - 2u)) : 0u;
      };
      for (uint64_t i = 0; exact && i + 1 < n; ++i)
        exact = guess(knots[i]) == std::min<uint64_t>(i, n - 2) &&
                guess(std::nextafter(knots[i + 1], knots[i])) == std::min<uint64_t>(i, n - 2);
    }
    guess_is_exact = exact;
  }

  // The bucket index is built on first use by a batch large enough to use it (>= 4096 queries): one-shot searches
  // and latency-bound small batches never pay for it.
  bool guess_is_exact = false;
  mutable std::once_flag lut_once;
  // A build that fails (out of device memory, a copy refused during stream capture) leaves the axis without the
  // index: the pyramid search serves every batch as before -- nothing is thrown out of call_once, no later call
  // retries, no evaluation fails because an optional accelerator could not be built.
  void ensure_bucket_index() const {   // lazily built cache: logically const
    std::call_once(lut_once, [this] {
      DevicePyramid* self = const_cast<DevicePyramid*>(this);
      try {
        maybe_fail_lazy_alloc(__LINE__);
        self->build_bucket_index(host_knots.data(), host_knots.size(), guess_is_exact);
      } catch (const HipFailure&) {
        (void)hipGetLastError();
        self->bidx = BucketIndex<T>{nullptr, 0, T(0)};
        self->lut_bytes = 0;
        self->lut_buf.release();
      }
    });
  }

  // The global-memory form for axes that are not staged in LDS (kernels.hpp, BucketIndex32): u32 entries, 2n <= m < 4n
  // buckets, at most 256 MiB.  Built on first use by a batch of >= 4096 queries.
  DevBuf lut32_buf;
  BucketIndex32<T> bidx32{nullptr, 0, T(0)};
  mutable std::once_flag lut32_once;
  void ensure_bucket_index32() const {
    std::call_once(lut32_once, [this] {
      DevicePyramid* self = const_cast<DevicePyramid*>(this);
      const T* knots = host_knots.data();
      const uint64_t n = host_knots.size();
      if (guess_is_exact || n <= 64 || n > (1ull << 25)) return;
      uint32_t m = 1;
      while (m < 2 * n) m *= 2;
      const T k0 = knots[0], kn = knots[n - 1];
      const T scale = T(m) / (kn - k0);
      if (!(scale > T(0)) || !std::isfinite((double)scale)) return;
      std::vector<uint32_t> lut((size_t)m + 2, 0);
      for (uint64_t i = 0; i < n; ++i) lut[bucket_of<T>(knots[i], k0, scale, m) + 1]++;   // counts, shifted by one
      for (uint32_t b = 0; b < m; ++b) lut[b + 1] += lut[b];                              // lut[b] = knots in buckets < b
      lut[m + 1] = (uint32_t)n;
      try {   // (failure: no index, the pyramid search from memory serves the axis -- see ensure_bucket_index)
        maybe_fail_lazy_alloc(__LINE__);
        self->lut32_buf.reserve(lut.size() * sizeof(uint32_t));
        NDI_HIP(hipMemcpy(self->lut32_buf.p, lut.data(), lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        self->bidx32 = BucketIndex32<T>{self->lut32_buf.template as<uint32_t>(), m, scale};
      } catch (const HipFailure&) {
        (void)hipGetLastError();
        self->bidx32 = BucketIndex32<T>{nullptr, 0, T(0)};
        self->lut32_buf.release();
      }
    });
  }

  // Dense bucket index of the query-per-lane kernels (kernels.hpp, DenseLut): so many uniform buckets that none holds
  // more than `maxk` knots -- the device then counts lut[bucket] + (the next maxk knots <= x) without a loop whose trip
  // count depends on the data.  4n buckets to start with, doubled until maxk <= 3 or the index would outgrow 64 KiB of
  // LDS; accepted up to maxk = 8 (clustered axes beyond that keep the other kernels).  Axes whose formula guess is exact
  // need no index.  Built on first use with the device's arithmetic; a failed build leaves dense_ok = false.
  DevBuf dlut_buf;
  DenseLut<T> dlut{nullptr, 0, 0, T(0)};
  size_t dlut_bytes = 0;   // LDS bytes of the staged index (16-byte multiple); 0 = none
  bool dense_ok = false;
  mutable std::once_flag dlut_once;
  void ensure_dense_lut() const {
    std::call_once(dlut_once, [this] {
      DevicePyramid* self = const_cast<DevicePyramid*>(this);
      const T* knots = host_knots.data();
      const uint64_t n = host_knots.size();
      if (n < 2) return;
      if (guess_is_exact) { self->dense_ok = true; return; }
      if (n > 16384) return;
      const T k0 = knots[0], kn = knots[n - 1];
      uint32_t m = 64;
      while (m < 4 * n) m *= 2;
      std::vector<uint32_t> cnt;
      // candidates m = 4n .. 32768 (64 KiB of LDS): the smallest index with one knot per bucket if that costs at most
      // 16 KiB, else the smallest with <= 3, else the densest (accepted up to 8)
      uint32_t best_m = 0, best_k = 0;
      T best_scale = T(0);
      uint32_t m1 = 0, m3 = 0, ml = 0, kl = 0;
      T s1 = T(0), s3 = T(0), sl = T(0);
      for (; m <= 32768; m *= 2) {
        const T scale = T(m) / (kn - k0);
        if (!(scale > T(0)) || !std::isfinite((double)scale)) break;
        cnt.assign(m, 0);
        uint32_t mk = 0;
        for (uint64_t i = 0; i < n; ++i) {
          T f = (knots[i] - k0) * scale;
          f = std::fmax(f, T(0));
          f = std::fmin(f, T(m - 1u));
          mk = std::max(mk, ++cnt[(uint32_t)f]);
        }
        ml = m; kl = mk; sl = scale;
        if (!m3 && mk <= 3) { m3 = m; s3 = scale; }
        if (!m1 && mk <= 1 && m <= 8192) { m1 = m; s1 = scale; }
        if (m1 || (m3 && m >= 8192)) break;
      }
      if (m1) { best_m = m1; best_k = 1; best_scale = s1; }
      else if (m3) { best_m = m3; best_k = 3; best_scale = s3; }
      else { best_m = ml; best_k = kl; best_scale = sl; }
      if (!best_m || best_k > 8) return;
      const T scale = best_scale;
      m = best_m;
      cnt.assign(m, 0);
      best_k = 0;
      for (uint64_t i = 0; i < n; ++i) {
        T f = (knots[i] - k0) * scale;
        f = std::fmax(f, T(0));
        f = std::fmin(f, T(m - 1u));
        best_k = std::max(best_k, ++cnt[(uint32_t)f]);
      }
      std::vector<uint16_t> lut((size_t)m + 2 + 6, (uint16_t)n);
      uint32_t run = 0;
      for (uint32_t b = 0; b < m; ++b) {
        lut[b] = (uint16_t)run;
        run += cnt[b];
      }
      const size_t bytes = (((size_t)(m + 2) / 2) * 4 + 15) & ~(size_t)15;
      try {
        maybe_fail_lazy_alloc(__LINE__);
        self->dlut_buf.reserve(std::max(bytes, lut.size() * sizeof(uint16_t)));
        NDI_HIP(hipMemcpy(self->dlut_buf.p, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      } catch (const HipFailure&) {
        (void)hipGetLastError();
        self->dlut_buf.release();
        return;
      }
      self->dlut = DenseLut<T>{self->dlut_buf.template as<uint16_t>(), m, best_k, scale};
      self->dlut_bytes = bytes;
      self->dense_ok = true;
    });
  }

  // Bucket index (kernels.hpp, BucketIndex): only for axes the O(1) formula guess does not resolve for every x, with
  // u16 entries (n <= 65535) and more than one top-level block.  Built with bucket_of(), the function the device uses.
  DevBuf lut_buf;
  BucketIndex<T> bidx{nullptr, 0, T(0)};
  size_t lut_bytes = 0;   // LDS bytes of the staged lut (16-byte multiple); 0 = none
  void build_bucket_index(const T* knots, uint64_t n, bool guess_is_exact) {
    bidx = BucketIndex<T>{nullptr, 0, T(0)};
    lut_bytes = 0;
    if (guess_is_exact || n <= 64 || n > 65535) return;
    uint32_t m = 1;
    while (m < 2 * n) m *= 2;                       // 2n <= m < 4n buckets
    const T k0 = knots[0], kn = knots[n - 1];
    const T scale = T(m) / (kn - k0);
    if (!(scale > T(0)) || !std::isfinite((double)scale)) return;   // degenerate span: keep the pyramid search
    std::vector<uint16_t> lut(m + 2 + 6, 0);        // m + 1 entries, padded to a whole number of 16-byte units
    std::vector<uint32_t> cnt(m, 0);
    for (uint64_t i = 0; i < n; ++i) cnt[bucket_of<T>(knots[i], k0, scale, m)]++;
    uint32_t run = 0;
    for (uint32_t b = 0; b < m; ++b) {
      lut[b] = (uint16_t)run;
      run += cnt[b];
    }
    for (size_t b = m; b < lut.size(); ++b) lut[b] = (uint16_t)n;   // lut[m] = n; padding likewise
    const size_t bytes = (((size_t)(m + 2) / 2) * 4 + 15) & ~(size_t)15;
    lut_buf.reserve(std::max(bytes, lut.size() * sizeof(uint16_t)));
    NDI_HIP(hipMemcpy(lut_buf.p, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    bidx = BucketIndex<T>{lut_buf.as<uint16_t>(), m, scale};
    lut_bytes = bytes;
  }
};

constexpr uint64_t MAX_KNOTS = (1ull << 31) - 1;          // interval indices are 32-bit
constexpr size_t LDS_STAGE_LIMIT = 150 * 1024;          // of the CU's 160 KiB
constexpr size_t FUSED_LDS_LIMIT = 160 * 1024;          // eval_fused_kernel with the tables in LDS: all of it

// ---------------------------------------------------------------------------------------------
// workspace: per (handle, stream) scratch, reused across stream-ordered evaluations
// ---------------------------------------------------------------------------------------------
// One set of per-chunk scratch: what the locate / group kernels of a chunk write and its evaluation reads.  A
// workspace carries two, so that the ring evaluation can locate + group chunk k+1 on a side stream while chunk k is
// being evaluated (the set is handed back by the eval_done event).
struct Scratch {
  DevBuf idx, idx2, t, perm, counts, cursor, hist, status, recq, chunkbin;
  DevBuf perm2, recq2, chist, cursor2;   // two-level 2-D grouping: coarse-ordered records, row histograms, consumable cursors
  hipEvent_t prep_done = nullptr, eval_done = nullptr;
  Scratch() = default;
  Scratch(const Scratch&) = delete;
  Scratch& operator=(const Scratch&) = delete;
  ~Scratch() {
    if (prep_done) (void)hipEventDestroy(prep_done);
    if (eval_done) (void)hipEventDestroy(eval_done);
  }
  void ensure_events() {
    if (!prep_done) NDI_HIP(hipEventCreateWithFlags(&prep_done, hipEventDisableTiming));
    if (!eval_done) NDI_HIP(hipEventCreateWithFlags(&eval_done, hipEventDisableTiming));
  }
};

struct Workspace {
  Scratch sc[2];
  DevBuf status, qdev, qdev2, stage;   // status: the range pre-pass of the ring / sharded evaluations
  StatusBlock* host_status = nullptr;  // pinned
  void* pin = nullptr;                 // pinned bounce buffer of the small-batch host path
  void* pin_dev = nullptr;             // its device-side address (hipHostGetDevicePointer)
  size_t pin_bytes = 0;
  hipStream_t side = nullptr;          // ring evaluation: locate + group of the next chunk run here
  hipEvent_t ev = nullptr;             // general-purpose ordering event (timing disabled)
  // record of the last batch (for finish())
  uint64_t last_nq = 0;
  const void* last_q = nullptr;
  const void* last_q2 = nullptr;
  int last_q_space = NDI_MEM_DEVICE;
  bool pending = false;
  // SpaceSet bookkeeping (guarded by the set's mutex)
  int users = 0;
  uint64_t last_use = 0;
  ~Workspace() {
    if (side) (void)hipStreamDestroy(side);   // waits for the stream's work
    if (ev) (void)hipEventDestroy(ev);
    if (host_status) (void)hipHostFree(host_status);
    if (pin) (void)hipHostFree(pin);
  }
  void ensure_pin(size_t need) {
    if (need <= pin_bytes) return;
    if (pin) (void)hipHostFree(pin);
    pin = nullptr;
    pin_dev = nullptr;
    pin_bytes = 0;
    NDI_HIP(hipHostMalloc(&pin, need, hipHostMallocMapped | hipHostMallocPortable));
    NDI_HIP(hipHostGetDevicePointer(&pin_dev, pin, 0));   // the address kernels use for the zero-copy path
    pin_bytes = need;
  }
  template <class U>
  U* pin_device(const void* host_ptr) const {   // device view of an address inside the pinned buffer
    return reinterpret_cast<U*>((char*)pin_dev + ((const char*)host_ptr - (const char*)pin));
  }
  void ensure_status() {
    status.reserve(sizeof(StatusBlock));
    for (Scratch& s : sc) s.status.reserve(sizeof(StatusBlock));
    if (!host_status) NDI_HIP(hipHostMalloc((void**)&host_status, sizeof(StatusBlock), hipHostMallocDefault));
  }
  hipStream_t side_stream() {
    if (!side) {   // highest priority: its small kernels are dispatched ahead of the evaluation's next workgroups
      int lo = 0, hi = 0;
      NDI_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      NDI_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    }
    return side;
  }
  hipEvent_t order_event() {
    if (!ev) NDI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return ev;
  }
};

// The scratch sets of one handle.  A set is *idle* when no call is inside it and its last batch has been
// collected (its stream was synchronised after the last kernel that touched the scratch).  At most MAX_IDLE idle
// sets are kept -- thread pools and short-lived streams would otherwise grow the map without bound; the least
// recently used idle set is freed first (hipFree / hipHostFree synchronise, so freeing is safe in any case).
struct SpaceSet {
  static constexpr size_t MAX_IDLE = 16;
  std::mutex mu;
  std::map<SpaceKey, std::unique_ptr<Workspace>> map;
  uint64_t tick = 0;

  Workspace& acquire(hipStream_t s) {
    std::lock_guard<std::mutex> g(mu);
    auto& slot = map[SpaceKey{s, std::this_thread::get_id()}];
    if (!slot) slot.reset(new Workspace());
    slot->users++;
    slot->last_use = ++tick;
    evict_locked(MAX_IDLE);
    return *slot;
  }
  void release(Workspace& w) {
    std::lock_guard<std::mutex> g(mu);
    w.users--;
  }
  void evict_locked(size_t keep) {
    for (;;) {
      size_t idle = 0;
      auto victim = map.end();
      for (auto it = map.begin(); it != map.end(); ++it) {
        if (it->second->users != 0 || it->second->pending) continue;
        ++idle;
        if (victim == map.end() || it->second->last_use < victim->second->last_use) victim = it;
      }
      if (idle <= keep || victim == map.end()) return;
      map.erase(victim);
    }
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    evict_locked(0);
  }
  size_t size() {
    std::lock_guard<std::mutex> g(mu);
    return map.size();
  }
};

struct SpaceLease {
  SpaceSet& set;
  Workspace& ws;
  SpaceLease(SpaceSet& st, hipStream_t s) : set(st), ws(st.acquire(s)) {}
  ~SpaceLease() { set.release(ws); }
  SpaceLease(const SpaceLease&) = delete;
  SpaceLease& operator=(const SpaceLease&) = delete;
};

// A ring of device buffers owned by a handle (ndi_ring_desc::slots == NULL): ONE allocation in which the slots
// are interleaved row by row -- row r of slot s lives at (r * n_slots + s) * row_stride, i.e. the chunk's rows have
// the pitch n_slots * row_stride.  Measured on MI355X (profiles/r02_placement_*.jsonl): the rate at
// which a kernel streams into a 32.8 GB extent depends on where the extent lies in physical memory (slots laid
// out one after the other in one allocation ran 6.1 / 5.9 / 5.6 / 4.7 ms for the same chunk, stable, and the same
// for sequential fills); with the rows of every slot striped over the whole ring each chunk's stream covers the
// ring's full extent and runs at the fast end (4.57-4.77 ms in 8 of 9 processes, < 1 % apart within a ring).
struct OwnedRing {
  std::mutex mu;   // one ring evaluation at a time uses the library-owned ring
  DevBuf buf;
  void ensure(uint32_t n, size_t chunk_rows, size_t stride_bytes) { buf.reserve((size_t)n * chunk_rows * stride_bytes); }
  void clear() { buf.release(); }
};

template <class T, class K, class A>
static void launch1(hipStream_t s, int cat, dim3 grid, dim3 block, size_t shmem, K kernel, const A& args) {
  ProfScope ps(s, cat);
  hipLaunchKernelGGL(kernel, grid, block, shmem, s, args);
  NDI_HIP(hipGetLastError());
  ps.done();
}

static void reset_status(void* status, hipStream_t s) {
  // first_fail[0..1] = NO_FAIL (all ones), the rest zero -- ONE launch (two hipMemsetAsync calls cost the host twice
  // the enqueue time of a kernel, and the host path of a 0.8 ms step matters: DESIGN.md 4.4)
  hipLaunchKernelGGL(reset_status_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<StatusBlock*>(status));
  NDI_HIP(hipGetLastError());
}

constexpr uint32_t BUCKETED_RUN = 1;         // consecutive 128-query chunks per workgroup of eval_bucketed_kernel
                                             // (1 / 4 / 8 / 16 / 32 measured equal within 1 %: tools/sweep_target.py)
constexpr uint32_t GROUP_MAX_BLOCKS = 256;   // query slices of the block-local counting sort (1-D; 2-D default)
constexpr uint32_t GROUP_MAX_BLOCKS_2D = 1024;   // upper limit of NDI_GROUP_BLOCKS for the 2-D tile grouping
// NDI_GROUP_BLOCKS (A/B, read once): fewer, longer slices -- longer runs per (slice, bin) in the record scatter
static uint32_t group_blocks() {
  static const uint32_t v = [] {
    const char* e = std::getenv("NDI_GROUP_BLOCKS");
    const int k = e ? std::atoi(e) : 0;
    return (uint32_t)(k >= 8 && k <= (int)GROUP_MAX_BLOCKS_2D ? k : (int)GROUP_MAX_BLOCKS);
  }();
  return v;
}
constexpr uint32_t GROUP_MAX_BINS = 16384;   // histogram must fit LDS next to the pyramid

template <class T>
static bool lds_sort_fits(const DevicePyramid<T>& pyr, uint64_t nb) {
  const size_t stage = pyr.lds_bytes <= LDS_STAGE_LIMIT ? ((pyr.lds_bytes + 15) & ~(size_t)15) : 0;
  return nb <= GROUP_MAX_BINS && stage + nb * 4 <= LDS_STAGE_LIMIT;   // (the bucket index is dropped first)
}

// Compute units of the current device (cached per device).
static unsigned cu_count() {
  static std::mutex mu;
  static std::map<int, unsigned> cache;
  int dev = 0;
  NDI_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(dev);
  if (it != cache.end()) return it->second;
  int n = 0;
  NDI_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
  return cache[dev] = (unsigned)std::max(n, 1);
}

// Workgroup size for a kernel that needs `lds` bytes per workgroup: as few threads as keep 32 waves on a CU
// (160 KiB LDS, at most 1024 threads per workgroup).
static unsigned threads_for_lds(size_t lds) {
  const size_t per_cu = std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
  unsigned best = 256;
  size_t best_waves = 0;
  for (unsigned threads : {256u, 512u, 1024u}) {
    const size_t waves_per_wg = threads / 64;
    const size_t wgs = std::min<size_t>(per_cu, 32 / waves_per_wg);
    const size_t waves = wgs * waves_per_wg;
    if (waves > best_waves) {
      best_waves = waves;
      best = threads;
    }
  }
  return best;
}

// Launches locate_kernel.  With `hist` the grid is one workgroup per contiguous query slice and each
// leaves its interval histogram in hist[b][nb]; *slice_out / *blocks_out describe the slicing.
template <class T>
static void run_locate(hipStream_t s, const DevicePyramid<T>& pyr, const T* q, uint64_t nq,
                       uint32_t* idx, int64_t* idx64, T* t, unsigned long long* first_fail, int mode,
                       uint32_t* hist = nullptr, uint32_t nb = 0, uint64_t* slice_out = nullptr,
                       uint32_t* blocks_out = nullptr, bool beside_eval = false) {
  LocateArgs<T> A{};
  A.pyr = pyr.view;
  A.q = q;
  A.nq = nq;
  A.idx = idx;
  A.idx64 = idx64;
  A.t = t;
  A.first_fail = first_fail;
  A.mode = mode;
  A.stage_lds = pyr.lds_bytes <= LDS_STAGE_LIMIT ? 1 : 0;
  size_t shmem = A.stage_lds ? ((pyr.lds_bytes + 15) & ~(size_t)15) : 0;
  uint64_t blocks = hist ? std::min<uint64_t>((nq + 2047) / 2048, GROUP_MAX_BLOCKS)
                         : std::min<uint64_t>((nq + 1023) / 1024, 2048);
  blocks = std::max<uint64_t>(blocks, 1);
  A.bx = BucketIndex<T>{nullptr, 0, T(0)};
  constexpr int lut_env = 1;   // (the bucket index: A/B settled in round 3)
  if (lut_env && A.stage_lds && nq >= 4096) pyr.ensure_bucket_index();
  if (lut_env && A.stage_lds && pyr.lut_bytes && nq >= 4096 &&
      shmem + pyr.lut_bytes + (hist ? (size_t)nb * 4 : 0) <= LDS_STAGE_LIMIT) {
    A.bx = pyr.bidx;          // bucket index staged behind the pyramid: [pyramid | lut | histogram]
    shmem += pyr.lut_bytes;
    // staging is the fixed cost of a workgroup now: no more workgroups than the chip holds at once
    const size_t total = shmem + (hist ? (size_t)nb * 4 : 0);
    blocks = std::min<uint64_t>(blocks, (uint64_t)cu_count() * std::max<size_t>(1, (160 * 1024) / total));
  }
  A.bx32 = BucketIndex32<T>{nullptr, 0, T(0)};
  if (lut_env && !A.stage_lds && nq >= 4096) {
    pyr.ensure_bucket_index32();
    A.bx32 = pyr.bidx32;
  }
  if (hist) shmem += (size_t)nb * 4;
  // beside a running evaluation kernel (ring pipeline) a 4-wave workgroup finds room wherever one evaluation
  // workgroup has retired; the 16-wave one that is best on an idle chip would wait for a quarter of a CU to drain
  const unsigned threads = beside_eval ? 256u : threads_for_lds(shmem);
  uint64_t slice = (nq + blocks - 1) / blocks;
  slice = (slice + threads - 1) / threads * threads;   // whole 64-query batches per wave
  blocks = (nq + slice - 1) / slice;
  A.slice = slice;
  A.hist = hist;
  A.nb = nb;
  allow_dynamic_lds(reinterpret_cast<const void*>(&locate_kernel<T, true, LOCATE_QB>), (int)LDS_STAGE_LIMIT);
  allow_dynamic_lds(reinterpret_cast<const void*>(&locate_kernel<T, false, LOCATE_QB>), (int)LDS_STAGE_LIMIT);
  allow_dynamic_lds(reinterpret_cast<const void*>(&locate_kernel<T, true, 1>), (int)LDS_STAGE_LIMIT);
  allow_dynamic_lds(reinterpret_cast<const void*>(&locate_kernel<T, false, 1>), (int)LDS_STAGE_LIMIT);
  if (slice_out) *slice_out = slice;
  if (blocks_out) *blocks_out = (uint32_t)blocks;
  const bool deep = slice / threads >= 16;   // batches per wave: queries of 4 batches requested together only then
  if (A.stage_lds && deep) launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate_kernel<T, true, LOCATE_QB>, A);
  else if (A.stage_lds) launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate_kernel<T, true, 1>, A);
  else if (deep) launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate_kernel<T, false, LOCATE_QB>, A);
  else launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate_kernel<T, false, 1>, A);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Tuning knobs of the short-row 1-D kernels (tools/short_rows_sweep.py).  Read once -- unless NDI_TUNE_LIVE is set
// when the library is loaded, in which case every evaluation re-reads them (one process then sweeps the variants).
//   NDI_SHORT_MODE   0 = auto, 1 = the two-kernel flat form (locate, then eval_flat_kernel), 2 = fused query order,
//                    3 = grouped (eval_bucketed_short_kernel)
//   NDI_FUSED_UNR    trips issued together by eval_fused_kernel (1 / 2 / 4)
//   NDI_FUSED_TB     its workgroup size (256 / 512 / 1024; 0 = chosen from the LDS footprint)
//   NDI_FUSED_LDS    tables in LDS: -1 = when they fit, 0 = never, 1 = whenever they fit (same as -1; kept for A/B)
//   NDI_FUSED_WGS    workgroups per CU of its grid (0 = default)
//   NDI_SHORT_CQ     grouped records per thread group and pass (16 / 64)
//   NDI_SHORT_ROWB   auto: the grouped form is taken from rows of this many bytes upwards
//   NDI_FUSED_PACK   interval-packed table copy for the fused kernel: -1 = rows shorter than a cache line,
//                    0 = never, 1 = always
// ndi_eval_opts as the entry points take it: defaults for NULL, unknown flag bits and a non-zero `reserved` refused (a
// caller compiled against an older header -- a shorter struct, uninitialised padding -- is told so instead of silently
// selecting an option), and NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED folded into the internal "no range pre-pass" bit.
static ndi_status take_opts(const ndi_eval_opts* opts, ndi_eval_opts& o) {
  o = ndi_eval_opts{};
  if (!opts) return NDI_OK;
  o = *opts;
  constexpr int32_t KNOWN = NDI_EVAL_FRESH_OUTPUT | NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED;
  if (o.flags & ~KNOWN) return fail(NDI_BAD_ARG, "ndi_eval_opts.flags has unknown bits (0x%x): built against another header version?", (unsigned)o.flags);
  if (o.reserved != 0) return fail(NDI_BAD_ARG, "ndi_eval_opts.reserved must be 0 (got %d): built against another header version?", (int)o.reserved);
  if (o.flags & NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED) o.flags |= NDI_EVAL_FRESH_OUTPUT;
  return NDI_OK;
}

struct ShortKnobs {
  int mode = 0, unr = 2, tb = 0, lds = -1, wgs = 0, cq = 64, rowb = 1024, pack = -1, maxlv = 256;
  static int env(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
  }
  static ShortKnobs read() {
    ShortKnobs k;
    k.mode = env("NDI_SHORT_MODE", k.mode);
    k.unr = env("NDI_FUSED_UNR", k.unr);
    k.tb = env("NDI_FUSED_TB", k.tb);
    k.lds = env("NDI_FUSED_LDS", k.lds);
    k.wgs = env("NDI_FUSED_WGS", k.wgs);
    k.cq = env("NDI_SHORT_CQ", k.cq);
    k.rowb = env("NDI_SHORT_ROWB", k.rowb);
    k.pack = env("NDI_FUSED_PACK", k.pack);
    return k;
  }
};
static ShortKnobs short_knobs() {
  static const bool live = std::getenv("NDI_TUNE_LIVE") != nullptr;
  static const ShortKnobs once = ShortKnobs::read();
  return live ? ShortKnobs::read() : once;
}

// The ring's locate + group of chunk k + 1 run on a side stream beside chunk k's evaluation (measured against running
// them on the evaluation stream: profiles/r03_ring_overlap.md).
static constexpr bool ring_overlap() { return true; }

static ndi_status check_ring_desc(const ndi_ring_desc* ring, uint64_t lanes, uint64_t* stride) {
  if (!ring || ring->n_slots == 0 || ring->chunk_queries == 0)
    return fail(NDI_BAD_ARG, "ring needs n_slots >= 1 and chunk_queries >= 1");
  *stride = ring->row_stride ? ring->row_stride : lanes;
  if (*stride < lanes) return fail(NDI_BAD_ARG, "ring row_stride (%llu) < lanes (%llu)",
                                   (unsigned long long)*stride, (unsigned long long)lanes);
  if (ring->slots)
    for (uint32_t i = 0; i < ring->n_slots; ++i)
      if (!ring->slots[i]) return fail(NDI_BAD_ARG, "ring slot %u is null", i);
  return NDI_OK;
}

// ---------------------------------------------------------------------------------------------
// Interp1D
// ---------------------------------------------------------------------------------------------
// NDI_BUILD_TIMING=1: wall-clock milestones of create() on stderr (where a build's milliseconds go: tools/build_probe.py)
struct BuildClock {
  bool on;
  std::chrono::steady_clock::time_point t0, last;
  BuildClock() : on(std::getenv("NDI_BUILD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), last(t0) {}
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ndi build] %-28s +%8.3f ms  (%8.3f)\n", what,
                 std::chrono::duration<double, std::milli>(now - last).count(),
                 std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

// Temporaries of small builds: one device buffer and one pinned staging buffer per host thread and device, grown on
// demand and kept (a build of (100, 5) spends more time in hipMalloc / hipFree / pageable copies than in its kernel).
// Deliberately never freed: thread-exit and process-exit order against the HIP runtime is not ours to rely on.
struct BuildScratch {
  std::map<int, std::pair<void*, size_t>> dev;   // device ordinal -> (buffer, bytes)
  void* pin = nullptr;
  size_t pin_bytes = 0;
  void* device_buf(int device, size_t need) {
    auto& e = dev[device];
    if (need > e.second) {
      if (e.first) (void)hipFree(e.first);
      e = {nullptr, 0};
      const size_t cap = std::max<size_t>(need, 64 * 1024);
      NDI_HIP(hipMalloc(&e.first, cap));
      e.second = cap;
    }
    return e.first;
  }
  void* pinned(size_t need) {
    if (need > pin_bytes) {
      if (pin) (void)hipHostFree(pin);
      pin = nullptr;
      pin_bytes = 0;
      const size_t cap = std::max<size_t>(need, 64 * 1024);
      NDI_HIP(hipHostMalloc(&pin, cap, hipHostMallocPortable));
      pin_bytes = cap;
    }
    return pin;
  }
};
static BuildScratch& build_scratch() {
  static thread_local BuildScratch* p = nullptr;
  if (!p) p = new BuildScratch();
  return *p;
}

// What makes two handles replicas of ONE interpolator beyond element type and lanes: knot count and values,
// strategy, extrapolation mode.  (Data and coefficient tables live on the devices and are not compared.)
static uint64_t fnv1a(uint64_t h, const void* p, size_t bytes) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}
constexpr uint64_t FNV_SEED = 0xcbf29ce484222325ull;

struct Interp1DBase {
  virtual ~Interp1DBase() = default;
  virtual uint64_t signature() const = 0;
  int dtype = 0, device = 0;
  uint64_t lanes = 0;
  virtual ndi_status eval(const void* q, uint64_t nq, void* out, uint64_t out_stride,
                          const ndi_eval_opts* opts, ndi_oob_info* info) = 0;
  virtual ndi_status finish(void* stream, ndi_oob_info* info) = 0;
  virtual ndi_status coefficients(void* a_out, void* b_out, int memspace) = 0;
  virtual ndi_status eval_ring(const void* q, uint64_t nq, const ndi_ring_desc* ring, ndi_ring_consumer consume,
                               void* user, const ndi_eval_opts* opts, ndi_oob_info* info) = 0;
  virtual ndi_status trim() = 0;
  virtual uint64_t scratch_sets() = 0;
  virtual ndi_status clone_to(int dev, Interp1DBase** out) = 0;
};

template <class T>
struct Interp1DImpl final : Interp1DBase {
  int strategy = NDI_LINEAR;
  int mode = EX_NO;
  uint64_t n = 0;
  DevBuf arena;   // small handles: ONE allocation behind pyr.buf / data / ca / cb / ck (create1d); declared first: freed last
  DevicePyramid<T> pyr;
  DevBuf data, ca, cb;
  DevBuf ck;   // the spline's derivatives k [n][lanes]: kept by builds whose {y, k} fit LDS (eval_fused_kernel, TLDS == 2)
  SpaceSet spaces;
  OwnedRing ring_own;
  // interval-packed copy of the tables for rows shorter than a cache line (pack_intervals_kernel), built on first
  // use by a batch that takes the query-order kernel; a replica builds its own
  DevBuf packed;
  static constexpr size_t PACKED_LIMIT = 1ull << 30;
  // 0 = not tried, 1 = ready and known complete, 2 = unavailable (too large / allocation failed: the unpacked tables serve
  // every batch), 3 = built by a kernel enqueued on packed_stream, completion signalled by packed_ev
  std::atomic<int> packed_state{0};
  std::mutex packed_mu;
  hipEvent_t packed_ev = nullptr;
  hipStream_t packed_stream = nullptr;
  ~Interp1DImpl() override {
    if (packed_ev) (void)hipEventDestroy(packed_ev);
  }
  // The copy is made by a kernel on the CALLER's stream (no NULL-stream launch, no device-wide synchronisation on the
  // evaluation path: the ring's side stream keeps running); calls on other streams are ordered behind it with an event
  // until it is known complete.  Never while the stream is being captured, and never fatal: a failed allocation leaves
  // the handle on the unpacked tables.
  bool ensure_packed(hipStream_t s) {
    int st = packed_state.load(std::memory_order_acquire);
    if (st == 1) return true;
    if (st == 2) return false;
    std::lock_guard<std::mutex> g(packed_mu);
    st = packed_state.load(std::memory_order_acquire);
    if (st == 1) return true;
    if (st == 2) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
    if (cs != hipStreamCaptureStatusNone) return false;   // this batch reads the unpacked tables
    if (st == 0) {
      const int parts = strategy == NDI_CUBIC_SPLINE ? 4 : 2;
      const size_t bytes = (size_t)(n - 1) * parts * lanes * sizeof(T);
      if (bytes == 0 || bytes > PACKED_LIMIT) {
        packed_state.store(2, std::memory_order_release);
        return false;
      }
      try {
        maybe_fail_lazy_alloc(__LINE__);
        packed.reserve(bytes);
        if (!packed_ev) NDI_HIP(hipEventCreateWithFlags(&packed_ev, hipEventDisableTiming));
        const uint64_t total = (n - 1) * (uint64_t)parts * lanes;
        const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((total + BLOCK - 1) / BLOCK, 65536));
        hipLaunchKernelGGL(pack_intervals_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, (const T*)data.as<T>(),
                           (const T*)ca.as<T>(), (const T*)cb.as<T>(), packed.as<T>(), n, lanes, parts);
        NDI_HIP(hipGetLastError());
        NDI_HIP(hipEventRecord(packed_ev, s));
      } catch (const HipFailure&) {
        (void)hipGetLastError();
        packed.release();
        packed_state.store(2, std::memory_order_release);
        return false;
      }
      packed_stream = s;
      packed_state.store(3, std::memory_order_release);
      return true;
    }
    // st == 3
    if (hipEventQuery(packed_ev) == hipSuccess) {
      packed_state.store(1, std::memory_order_release);
      return true;
    }
    (void)hipGetLastError();
    if (s != packed_stream) NDI_HIP(hipStreamWaitEvent(s, packed_ev, 0));
    return true;
  }

  uint64_t signature() const override {
    uint64_t h = fnv1a(FNV_SEED, pyr.host_knots.data(), pyr.host_knots.size() * sizeof(T));
    const uint64_t f[3] = {n, (uint64_t)strategy, (uint64_t)mode};
    return fnv1a(h, f, sizeof(f));
  }

  // The derivatives k are kept next to a / b when {y, k} (2 n lanes elements) can sit in LDS beside the knots and the
  // smallest strip set: the query-order kernel then re-forms a / b per item instead of gathering them (TLDS == 2).
  // NDI_SPLINE_KEEP_K=0 drops them.
  T* reserve_k() {
    static const bool tune_live = std::getenv("NDI_TUNE_LIVE") != nullptr;
    static const int once = ShortKnobs::env("NDI_SPLINE_KEEP_K", 1);
    const int keep = tune_live ? ShortKnobs::env("NDI_SPLINE_KEEP_K", 1) : once;
    if (!keep || lanes > 2048 || fused_lds_bytes(false, 256, 2) > FUSED_LDS_LIMIT) {
      ck.release();   // (a slice of the handle's arena may have been set aside: ck.p != nullptr means "k was kept")
      return nullptr;
    }
    ck.reserve((size_t)n * lanes * sizeof(T));
    return ck.as<T>();
  }

  // ---- build (CubicSpline::build, cubic_spline.rs:754-771) --------------------------------
  ndi_status build_spline(const ndi_interp1d_desc& d) {
    Range rg("ndi:spline_build");
    const bool per_lane = d.lane_left_kind || d.lane_left_value || d.lane_right_kind || d.lane_right_value;
    if (per_lane) {
      if (!(d.lane_left_kind && d.lane_left_value && d.lane_right_kind && d.lane_right_value))
        return fail(NDI_BAD_ARG, "BoundaryCondition::Individual needs all four lane_* arrays");
      if (d.periodic) return fail(NDI_BAD_ARG, "Periodic cannot be combined with per-lane boundaries");
      return build_spline_individual(d);
    }
    const bool periodic = d.periodic != 0;
    BuildClock clk;
    const size_t tab = (size_t)(n - 1) * lanes * sizeof(T);
    ca.reserve(tab);
    cb.reserve(tab);
    // Blocked sweeps or serial kernels?  Narrow trailing axes with many knots: the per-lane serial kernel would be one
    // or two waves doing 2n dependent steps.  The blocked sweeps (kernels.hpp, spline_blocked_*) take over -- the one
    // path whose tables are not bit-identical to the reference order (a few ulp; NDI_SPLINE_BLOCKED=0 keeps the serial
    // kernels, =1 forces the blocked ones wherever they apply).
    static const bool tune_live = std::getenv("NDI_TUNE_LIVE") != nullptr;
    static const int blocked_once = ShortKnobs::env("NDI_SPLINE_BLOCKED", -1);
    const int blocked_env = tune_live ? ShortKnobs::env("NDI_SPLINE_BLOCKED", -1) : blocked_once;
    // ... and only on axes whose neighbouring knot spacings differ by less than 1e3 (f32) / 1e9 (f64): the re-associated
    // sweeps are accurate relative to the NEIGHBOURING table magnitudes, and where the spacing jumps by 1e6 from one
    // interval to the next those magnitudes jump likewise -- in f32 evaluated rows then miss the 1e-5 bar (4 of 5000 rows
    // at 4e-5 on gaps drawn log-uniformly over six decades, tests/test_gpu_spline_blocked.py).  Such axes keep the serial
    // kernels: bit-identical.
    bool tame = true;
    if (blocked_env != 0 && n >= 16) {      // (also when the blocked build is forced: the device-side elimination needs it)
      const double lim = sizeof(T) == 4 ? 1e3 : 1e9;
      const T* xs = pyr.host_knots.data();
      for (uint64_t i = 2; tame && i < n; ++i) {
        const double a = (double)xs[i] - (double)xs[i - 1], b = (double)xs[i - 1] - (double)xs[i - 2];
        tame = a <= lim * b && b <= lim * a;
      }
    }
    // (n >= 16: the system is the GENERAL or the PERIODIC one -- the closed forms are n == 3)
    const bool blocked = n >= 16 && !(d.build_flags & NDI_BUILD_REFERENCE_ORDER) &&
                         (blocked_env > 0 || (blocked_env < 0 && n >= 2048 && lanes <= 256 && tame));
    // Long axes: the x-only elimination factors on the device as well (spline_eliminate_kernel); the host then forms only
    // the boundary rows' scalars.
    constexpr int elim_env = 1;
    const bool dev_elim = blocked && tame && !periodic && n >= 32768 && elim_env != 0;
    SplineEnds<T> ends;
    SplinePlan<T> P = dev_elim ? make_spline_plan_scalars<T>(pyr.host_knots.data(), n, d.left.kind, d.left.value, d.right.kind,
                                                             d.right.value, ends)
                               : make_spline_plan<T>(pyr.host_knots.data(), n, periodic, d.left.kind, d.left.value,
                                                     d.right.kind, d.right.value);
    clk.mark("  host plan");
    // Small systems (the reference's (100, 5); 1024 x 8; ...): ONE launch -- right-hand sides, elimination, back
    // substitution and the a / b epilogue in spline_build_general_kernel<FUSED>, dx / up formed from the resident knots;
    // the x-only factors w, mid' travel through a kept pinned buffer into a kept device buffer: no allocation, no free,
    // no status read-back (only the periodic build has one).  Same operations as the three-kernel form: bit-identical.
    constexpr int small_env = 1;
    if (small_env && P.mode == SPLINE_GENERAL && !blocked && (uint64_t)n * lanes <= (1u << 17)) {
      BuildScratch& bs = build_scratch();
      const size_t plan_b = 2 * (size_t)n * sizeof(T);
      T* hp = static_cast<T*>(bs.pinned(plan_b));
      T* dp = static_cast<T*>(bs.device_buf(device, plan_b));
      std::memcpy(hp, P.w.data(), (size_t)n * sizeof(T));
      std::memcpy(hp + n, P.midp.data(), (size_t)n * sizeof(T));
      NDI_HIP(hipMemcpyAsync(dp, hp, plan_b, hipMemcpyHostToDevice, nullptr));
      clk.mark("    plan staged + copy enqueued");
      BuildArgs<T> A{};
      A.data = data.as<T>();
      A.ca = ca.as<T>();
      A.cb = cb.as<T>();
      A.x = pyr.view.lv0;
      A.w = dp;
      A.midp = dp + n;
      A.up_len = P.up.size();
      A.left_up0 = P.up.front();
      A.up_last = P.up.back();
      A.n = n;
      A.lanes = lanes;
      A.left_kind = P.left_kind;
      A.right_kind = P.right_kind;
      A.left_val = P.left_val;
      A.right_val = P.right_val;
      A.nkL_tmp1 = P.nkL_tmp1; A.nkL_d = P.nkL_d;
      A.nkR_tmp1 = P.nkR_tmp1; A.nkR_d = P.nkR_d;
      A.dx0_sq = P.dx0_sq; A.dxl_sq = P.dxl_sq;
      A.kout = reserve_k();
      const unsigned grid1 = (unsigned)((lanes + 63) / 64);
      // the smallest systems -- data and right-hand sides fit LDS twice over, one workgroup covers the trailing axis --
      // form their right-hand sides with the whole workgroup and sweep out of LDS (spline_build_lds_kernel)
      constexpr int lds_env = 1;
      const size_t lds_need = 2 * (size_t)n * lanes * sizeof(T);
      if (lds_env && lanes <= (uint64_t)BLOCK && lds_need <= 96 * 1024 && n >= 4) {
        allow_dynamic_lds(reinterpret_cast<const void*>(&spline_build_lds_kernel<T, true>), 96 * 1024);
        allow_dynamic_lds(reinterpret_cast<const void*>(&spline_build_lds_kernel<T, false>), 96 * 1024);
        if (A.kout) hipLaunchKernelGGL((spline_build_lds_kernel<T, true>), dim3(1), dim3(BLOCK), lds_need, (hipStream_t) nullptr, A);
        else hipLaunchKernelGGL((spline_build_lds_kernel<T, false>), dim3(1), dim3(BLOCK), lds_need, (hipStream_t) nullptr, A);
      } else if (A.kout) hipLaunchKernelGGL((spline_build_general_kernel<T, false, true, true>), dim3(grid1), dim3(64), 0, (hipStream_t) nullptr, A);
      else hipLaunchKernelGGL((spline_build_general_kernel<T, false, false, true>), dim3(grid1), dim3(64), 0, (hipStream_t) nullptr, A);
      NDI_HIP(hipGetLastError());
      clk.mark("    kernel enqueued");
      NDI_HIP(hipStreamSynchronize(nullptr));   // the tables are complete when create() returns: any stream may read them
      clk.mark("  fused small build");
      return NDI_OK;
    }
    const bool per = P.mode == SPLINE_PERIODIC;
    const uint64_t rows = per ? n - 2 : n;   // order of the system the two sweeps run over
    uint64_t S = 64;
    while (S * S < rows && S < 2048) S *= 2;         // ~sqrt(n) rows per block: local sweeps and carry chain balance
    const uint64_t nblk = (rows + S - 1) / S;
    // ONE temporary allocation (one hipMalloc, one hipFree -- each costs as much as the kernels of a small build):
    //   [status | dx (n) | up (n) | w (n) | mid' (n) | k2 (n) | blocked: fP | dco | bP | rfull | ends | carry]
    // dx and up are formed on the device from the knots already there (the same subtractions in T as the host plan's);
    // w, mid' and k2 -- the results of the host's division chain -- are uploaded straight from their vectors.
    const size_t plan_elems = 5 * (size_t)n;
    const size_t blk_elems = blocked ? 3 * (size_t)n + (size_t)n * lanes + 2 * (size_t)nblk * lanes : 0;
    // temporaries: kept per host thread up to 64 MiB (hipMalloc + hipFree of a 48 MB buffer cost 4 ms of a 1e6-knot build,
    // and more than the kernels of a 4096 x 8 one); larger ones are allocated and freed here
    DevBuf tmp;
    const size_t tmp_bytes = 256 + (plan_elems + blk_elems) * sizeof(T);
    void* tmp_p = nullptr;
    if (tmp_bytes <= ((size_t)64 << 20)) {
      tmp_p = build_scratch().device_buf(device, tmp_bytes);
    } else {
      tmp.reserve(tmp_bytes);
      tmp_p = tmp.p;
    }
    NDI_HIP(hipMemsetAsync(tmp_p, 0, sizeof(StatusBlock), nullptr));
    T* const plan = reinterpret_cast<T*>((char*)tmp_p + 256);
    T* const d_dx = plan;
    T* const d_up = plan + n;
    T* const d_w = plan + 2 * (size_t)n;
    T* const d_mid = plan + 3 * (size_t)n;
    T* const d_k2 = plan + 4 * (size_t)n;
    if (!P.w.empty()) NDI_HIP(hipMemcpyAsync(d_w, P.w.data(), P.w.size() * sizeof(T), hipMemcpyHostToDevice, nullptr));
    if (!P.midp.empty()) NDI_HIP(hipMemcpyAsync(d_mid, P.midp.data(), P.midp.size() * sizeof(T), hipMemcpyHostToDevice, nullptr));
    if (!P.k2.empty()) NDI_HIP(hipMemcpyAsync(d_k2, P.k2.data(), P.k2.size() * sizeof(T), hipMemcpyHostToDevice, nullptr));
    {
      const uint64_t up_len = dev_elim ? n : P.up.size();
      const T up_first = dev_elim ? ends.up_first : (up_len ? P.up[0] : T(0));
      const T up_last = dev_elim ? T(0) : (up_len ? P.up[up_len - 1] : T(0));
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 4096));
      hipLaunchKernelGGL(spline_dx_up_kernel<T>, dim3(g), dim3(BLOCK), 0, (hipStream_t) nullptr, (const T*)pyr.view.lv0, d_dx, d_up,
                         n, up_len, up_first, up_last);
      if (dev_elim) {   // w, mid' where the host plan would have been uploaded
        const uint64_t SE = 256;
        const uint64_t tasks = (n + SE - 1) / SE;
        hipLaunchKernelGGL(spline_eliminate_kernel<T>, dim3((unsigned)((tasks + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t) nullptr,
                           (const T*)pyr.view.lv0, n, ends.up_first, ends.mid_first, ends.low_last, ends.mid_last, d_w, d_mid, SE);
      }
      NDI_HIP(hipGetLastError());
    }
    clk.mark("  tables alloc + plan upload");

    BuildArgs<T> A{};
    A.data = data.as<T>();
    A.ca = ca.as<T>();
    A.cb = cb.as<T>();
    A.dx = d_dx;
    A.up = d_up;
    A.w = d_w;
    A.midp = d_mid;
    A.k2 = d_k2;
    A.n = n;
    A.lanes = lanes;
    A.left_kind = P.left_kind;
    A.right_kind = P.right_kind;
    A.left_val = P.left_val;
    A.right_val = P.right_val;
    A.nkL_tmp1 = P.nkL_tmp1; A.nkL_d = P.nkL_d;
    A.nkR_tmp1 = P.nkR_tmp1; A.nkR_d = P.nkR_d;
    A.dx0_sq = P.dx0_sq; A.dxl_sq = P.dxl_sq;
    A.per_den = P.per_den;
    A.status = static_cast<StatusBlock*>(tmp_p);
    A.kout = reserve_k();
    const unsigned grid = (unsigned)((lanes + 63) / 64);
    hipStream_t s = nullptr;
    if (blocked) {
      T* sp = plan + plan_elems;     // [fP | dco | bP | rfull | ends | carry]; the coefficient products are formed on the device
      A.fP = sp;
      A.dco = sp + n;
      A.bP = sp + 2 * n;
      A.rfull = sp + 3 * n;
      A.ends = A.rfull + n * lanes;
      A.carry = A.ends + nblk * lanes;
      A.S = S;
      A.nblocks = nblk;
      A.rows = rows;
      const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n * lanes + BLOCK - 1) / BLOCK, 65536));
      const unsigned gl = (unsigned)((nblk * lanes + BLOCK - 1) / BLOCK);
      hipLaunchKernelGGL(spline_blocked_coef_kernel<T>, dim3((unsigned)((nblk + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                         A.w, A.up, A.midp, sp, sp + n, sp + 2 * n, rows, S, nblk);
      if (per) hipLaunchKernelGGL(spline_periodic_rhs_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, A);
      else hipLaunchKernelGGL((spline_rhs_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, s, A);
      hipLaunchKernelGGL(spline_blocked_local_kernel<T>, dim3(gl), dim3(BLOCK), 0, s, A, 0);
      hipLaunchKernelGGL(spline_blocked_carry_kernel<T>, dim3(grid), dim3(64), 0, s, A, 0);
      hipLaunchKernelGGL(spline_blocked_fix_forward_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, A);
      hipLaunchKernelGGL(spline_blocked_local_kernel<T>, dim3(gl), dim3(BLOCK), 0, s, A, 1);
      hipLaunchKernelGGL(spline_blocked_carry_kernel<T>, dim3(grid), dim3(64), 0, s, A, 1);
      if (per) {
        hipLaunchKernelGGL(spline_periodic_km1_kernel<T>, dim3(grid), dim3(64), 0, s, A);
        hipLaunchKernelGGL(spline_periodic_finish_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, A);
      } else {
        hipLaunchKernelGGL(spline_blocked_finish_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, A);
      }
      NDI_HIP(hipGetLastError());
      StatusBlock hs{};
      if (per) NDI_HIP(hipMemcpy(&hs, tmp_p, sizeof(hs), hipMemcpyDeviceToHost));   // (only the periodic build reports through it)
      else NDI_HIP(hipStreamSynchronize(s));                                         // the tables are complete on return
      clk.mark("  blocked sweeps");
      if (hs.periodic_mismatch != 0)
        return fail(NDI_VALUE,
                    "for periodic boundary condition the first and last value must be equal "
                    "(%llu lane(s) differ)", hs.periodic_mismatch);
      if (mode != EX_NO && periodic) mode = EX_PERIODIC;
      return NDI_OK;
    }
    switch (P.mode) {
      case SPLINE_GENERAL: {
        // Wide trailing axes: four waves per 64 lanes, the right-hand sides never leave the chip (spline_build_wide_kernel).
        // NDI_SPLINE_WIDE=0 / 1: A/B (1 takes it for every width).
        static const bool tune_live_w = std::getenv("NDI_TUNE_LIVE") != nullptr;
        static const int wide_once = ShortKnobs::env("NDI_SPLINE_WIDE", -1);
        const int wide_env = tune_live_w ? ShortKnobs::env("NDI_SPLINE_WIDE", -1) : wide_once;
        if (!A.kout && n >= 4 && (wide_env > 0 || (wide_env < 0 && lanes >= 1024))) {
          constexpr int RW = 16, RBW = 3 * RW;       // rows per producer wave / per block
          const size_t shm = (size_t)(4 * RBW * 64 + 2 * 4 * RBW) * sizeof(T);
          auto kern = spline_build_wide_kernel<T, RW>;
          allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)shm);
          hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, s, A);
          break;
        }
        const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n * lanes + BLOCK - 1) / BLOCK, 65536));
        hipLaunchKernelGGL((spline_rhs_kernel<T, false>), dim3(gr), dim3(BLOCK), 0, s, A);
        if (A.kout) hipLaunchKernelGGL((spline_build_general_kernel<T, false, true>), dim3(grid), dim3(64), 0, s, A);
        else hipLaunchKernelGGL((spline_build_general_kernel<T, false, false>), dim3(grid), dim3(64), 0, s, A);
        break;
      }
      case SPLINE_PARABOLA3:
        hipLaunchKernelGGL(spline_build_n3_kernel<T>, dim3(grid), dim3(64), 0, s, A, 0);
        break;
      case SPLINE_PERIODIC3:
        hipLaunchKernelGGL(spline_build_n3_kernel<T>, dim3(grid), dim3(64), 0, s, A, 1);
        break;
      case SPLINE_PERIODIC:
        hipLaunchKernelGGL(spline_build_periodic_kernel<T>, dim3(grid), dim3(64), 0, s, A);
        break;
    }
    NDI_HIP(hipGetLastError());
    StatusBlock hs{};
    NDI_HIP(hipMemcpy(&hs, tmp_p, sizeof(hs), hipMemcpyDeviceToHost));  // synchronises
    if (hs.periodic_mismatch != 0)
      return fail(NDI_VALUE,
                  "for periodic boundary condition the first and last value must be equal "
                  "(%llu lane(s) differ)", hs.periodic_mismatch);
    // Extrapolate::{No,Yes,Periodic}, cubic_spline.rs:763-769
    if (mode != EX_NO && periodic) mode = EX_PERIODIC;
    return NDI_OK;
  }

  // BoundaryCondition::Individual (cubic_spline.rs:332-347 + solve_for_k_individual :370-403): one scalar
  // solve per trailing element in the reference; here every lane picks its end kinds / values and one of
  // the precomputed elimination plans (4 left kinds x 4 right kinds, kind 3 = the n == 3 parabola rows).
  ndi_status build_spline_individual(const ndi_interp1d_desc& d) {
    const T* x = pyr.host_knots.data();
    std::vector<uint8_t> cls(lanes);
    std::vector<T> lval(lanes), rval(lanes);
    for (uint64_t l = 0; l < lanes; ++l) {
      int lk, rk;
      double lv, rv;
      specialize_end(d.lane_left_kind[l], d.lane_left_value[l], lk, lv);
      specialize_end(d.lane_right_kind[l], d.lane_right_value[l], rk, rv);
      if (n == 3 && lk == END_NOT_A_KNOT && rk == END_NOT_A_KNOT) lk = rk = 3;  // :569-596
      cls[l] = (uint8_t)(lk | (rk << 2));
      lval[l] = T(lv);
      rval[l] = T(rv);
    }
    static const int to_ndi[3] = {NDI_BC_NOT_A_KNOT, NDI_BC_FIRST_DERIV, NDI_BC_SECOND_DERIV};
    std::vector<T> w4(4 * n, T(0)), midp4(4 * n, T(1)), up0_4(4, T(0)), wl(16, T(0)), midl(16, T(1));
    SplinePlan<T> ref;
    for (int lk = 0; lk < 3; ++lk)
      for (int rk = 0; rk < 3; ++rk) {
        SplinePlan<T> P = make_spline_plan<T>(x, n, false, to_ndi[lk], 0.0, to_ndi[rk], 0.0);
        int a = lk, b = rk;
        if (P.mode == SPLINE_PARABOLA3) a = b = 3;
        for (uint64_t i = 0; i + 1 < n; ++i) {
          w4[a * n + i] = P.w[i];
          midp4[a * n + i] = P.midp[i];
        }
        up0_4[a] = P.up[0];
        wl[a * 4 + b] = P.w[n - 1];
        midl[a * 4 + b] = P.midp[n - 1];
        if (lk == 2 && rk == 2) ref = P;  // a general-mode plan: shared dx / interior up / not-a-knot scalars
        if (P.mode != SPLINE_PARABOLA3 && lk == 0) { ref.nkL_tmp1 = P.nkL_tmp1; ref.nkL_d = P.nkL_d; }
        if (P.mode != SPLINE_PARABOLA3 && rk == 0) { ref.nkR_tmp1 = P.nkR_tmp1; ref.nkR_d = P.nkR_d; }
      }
    {  // not-a-knot row scalars depend on x only; take them from plans that have them (n == 3 included)
      SplinePlan<T> PL = make_spline_plan<T>(x, n, false, NDI_BC_NOT_A_KNOT, 0.0, NDI_BC_FIRST_DERIV, 0.0);
      SplinePlan<T> PR = make_spline_plan<T>(x, n, false, NDI_BC_FIRST_DERIV, 0.0, NDI_BC_NOT_A_KNOT, 0.0);
      ref.nkL_tmp1 = PL.nkL_tmp1; ref.nkL_d = PL.nkL_d;
      ref.nkR_tmp1 = PR.nkR_tmp1; ref.nkR_d = PR.nkR_d;
    }
    const size_t tab = (size_t)(n - 1) * lanes * sizeof(T);
    ca.reserve(tab);
    cb.reserve(tab);
    std::vector<T> pack;
    auto put = [&pack](const std::vector<T>& v) { size_t o = pack.size(); pack.insert(pack.end(), v.begin(), v.end()); return o; };
    const size_t o_dx = put(ref.dx), o_up = put(ref.up), o_w4 = put(w4), o_m4 = put(midp4), o_u0 = put(up0_4),
                 o_wl = put(wl), o_ml = put(midl), o_lv = put(lval), o_rv = put(rval);
    DevBuf plan, dcls;
    plan.reserve(pack.size() * sizeof(T));
    NDI_HIP(hipMemcpy(plan.p, pack.data(), pack.size() * sizeof(T), hipMemcpyHostToDevice));
    dcls.reserve(lanes);
    NDI_HIP(hipMemcpy(dcls.p, cls.data(), lanes, hipMemcpyHostToDevice));
    BuildArgs<T> A{};
    A.data = data.as<T>();
    A.ca = ca.as<T>();
    A.cb = cb.as<T>();
    const T* base = plan.as<T>();
    A.dx = base + o_dx; A.up = base + o_up; A.w = base + o_w4; A.midp = base + o_m4;
    A.w4 = base + o_w4; A.midp4 = base + o_m4; A.up0_4 = base + o_u0; A.wl = base + o_wl; A.midl = base + o_ml;
    A.lane_lval = base + o_lv; A.lane_rval = base + o_rv;
    A.lane_cls = dcls.as<uint8_t>();
    A.n = n;
    A.lanes = lanes;
    A.nkL_tmp1 = ref.nkL_tmp1; A.nkL_d = ref.nkL_d; A.nkR_tmp1 = ref.nkR_tmp1; A.nkR_d = ref.nkR_d;
    A.dx0_sq = ref.dx0_sq; A.dxl_sq = ref.dxl_sq;
    const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n * lanes + BLOCK - 1) / BLOCK, 65536));
    hipLaunchKernelGGL((spline_rhs_kernel<T, true>), dim3(gr), dim3(BLOCK), 0, (hipStream_t) nullptr, A);
    A.kout = reserve_k();
    if (A.kout)
      hipLaunchKernelGGL((spline_build_general_kernel<T, true, true>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0,
                         (hipStream_t) nullptr, A);
    else
      hipLaunchKernelGGL((spline_build_general_kernel<T, true, false>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0,
                         (hipStream_t) nullptr, A);
    NDI_HIP(hipGetLastError());
    NDI_HIP(hipDeviceSynchronize());
    return NDI_OK;
  }

  // ---- evaluation core on device pointers --------------------------------------------------
  // A batch is evaluated in two stages that may run on different streams: prep() = search (+ grouping) into a
  // scratch set, launch_eval() = the evaluation kernel reading that set.  Plan1 carries what prep() decided.
  struct Plan1 {
    enum Kind { SMALL, BUCKETED, ROWS, FLAT, FUSED, BUCKETED_SHORT, LANES } kind = ROWS;
    const T* q = nullptr;
    uint64_t nq = 0;
    T* out = nullptr;
    uint64_t out_stride = 0;
    bool vec_ok = false;
    uint64_t LV = 0;
    // FUSED (eval_fused_kernel)
    bool f_lut = false, f_pack = false;
    bool f_sorted = false;   // FUSED: the queries of a workgroup round ordered by interval in LDS (eval_fused_sorted_kernel)
    int f_tlds = 0;   // 0: tables from memory, 1: {y, a, b} in LDS, 2: {y, k} in LDS
    unsigned f_tb = 256, f_grid = 1;
    int f_unr = 2;
    size_t f_lds = 0;
    int s_cq = 64;   // BUCKETED_SHORT
    int l_qpl = 1;   // LANES, scalar data: queries per lane (1, or one 16-byte vector)
    bool l_check = false;   // LANES: no range pre-pass, the kernel checks the queries itself (NDI_EVAL_FRESH_OUTPUT)
  };

  // LDS footprint of eval_scalar_kernel / eval_lanes_kernel: [knots | dense index | interval records | table records | strips]
  size_t lanes_lds_bytes(unsigned tb) const {
    const size_t tr = strategy == NDI_CUBIC_SPLINE ? 4 : std::max<size_t>(2, 16 / sizeof(T));   // (whole 16-byte units)
    size_t b = ((size_t)(n + LANE_SENTINELS) * sizeof(T) + 15) & ~(size_t)15;
    b += pyr.dlut_bytes;
    if (strategy == NDI_CUBIC_SPLINE) b += (size_t)(n - 1) * 4 * sizeof(T);
    b += ((size_t)(n - 1) * lanes * tr * sizeof(T) + 15) & ~(size_t)15;
    if (lanes > 1) b += (size_t)(tb / 64) * 64 * lanes * sizeof(T);
    return b;
  }

  // Query per lane with the whole table set in LDS (eval_scalar_kernel / eval_lanes_kernel): rows of up to 56 bytes
  // whose records fit LDS beside the knots, an axis the branch-free search covers (dense bucket index
  // or exact O(1) guess), batches that give every workgroup several times its staging bytes to write.
  // NDI_LANES_KERNEL=0 leaves these shapes to the query-order kernel (A/B); =1 takes it whenever it fits.
  bool plan_lanes(hipStream_t s, Scratch& sc, Plan1& P, int path, int flags) {
    static const bool tune_live = std::getenv("NDI_TUNE_LIVE") != nullptr;
    // rows of up to 56 bytes: at 64 bytes the query-order kernel is level (f64 x 8: 65 vs 61-65 Gqueries/s) or ahead
    // (f32 x 16: 74 vs 57), profiles/r05_small_shapes_rates.txt
    static const int on_once = ShortKnobs::env("NDI_LANES_KERNEL", -1);
    const int on = tune_live ? ShortKnobs::env("NDI_LANES_KERNEL", -1) : on_once;
    constexpr int maxb = 56;
    if (on == 0 || path == NDI_PATH_BUCKETED || n < 3 || n > 16384) return false;
    if (on < 0 && short_knobs().mode != 0) return false;   // a pinned short-row variant (NDI_SHORT_MODE) is what runs
    if (lanes * sizeof(T) > (size_t)(on > 0 ? std::max(maxb, 64) : maxb)) return false;   // (forced: up to 64 bytes, for the tests)
    const size_t tab = ((size_t)n * sizeof(T)) + (size_t)(n - 1) * (4 + lanes * 4) * sizeof(T);   // (before the index is known)
    if (tab > FUSED_LDS_LIMIT) return false;
    if (on < 0 && (P.nq < 65536 || (double)P.nq * (double)lanes * sizeof(T) < 4.0 * (double)cu_count() * (double)tab)) return false;
    pyr.ensure_dense_lut();
    if (!pyr.dense_ok) return false;
    // workgroup: 256 threads when four or more fit a CU beside each other (small tables: the staging pass is cheap and
    // short workgroups retire independently), else 1024
    P.f_tb = lanes_lds_bytes(256) * 4 <= 160 * 1024 ? 256u : 1024u;
    if (lanes_lds_bytes(P.f_tb) > FUSED_LDS_LIMIT) P.f_tb = 256u;   // (the strips of 16 waves do not fit: 4 waves)
    P.f_lds = lanes_lds_bytes(P.f_tb);
    if (P.f_lds > FUSED_LDS_LIMIT) return false;
    constexpr int VN = Wide<T>::N;
    P.l_qpl = (lanes == 1 && P.out_stride == 1 && aligned16(P.q) && aligned16(P.out)) ? VN : 1;
    const size_t wg_per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / P.f_lds, 32 / (P.f_tb / 64)));
    const uint64_t per_wg = (uint64_t)P.f_tb * (lanes == 1 ? (uint64_t)P.l_qpl : 1);
    P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + per_wg - 1) / per_wg, (uint64_t)cu_count() * wg_per_cu));
    P.kind = Plan1::LANES;
    g_last_path.store(NDI_PATH_GATHER);
    // fresh output (interp_array: the buffer is dropped on Err): no pre-pass, the kernel's own range test reports the failure
    P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;
    if (P.l_check) return true;
    StatusBlock* st = sc.status.as<StatusBlock>();
    const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + BLOCK - 1) / BLOCK, 4096));
    ProfScope ps(s, PC_LOCATE);
    hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, P.q, (const T*)nullptr, P.nq, k0, kn, k0, kn,
                       mode, &st->first_fail[0]);
    NDI_HIP(hipGetLastError());
    ps.done();
    return true;
  }

  void launch_lanes(hipStream_t s, Scratch& sc, const Plan1& P) {
    EvalLanesArgs<T> F{};
    F.knots = pyr.view.lv0;
    F.n = (uint32_t)n;
    F.dl = pyr.dlut;
    F.data = data.as<T>();
    F.ca = ca.as<T>();
    F.cb = cb.as<T>();
    F.q = P.q;
    F.out = P.out;
    F.nq = P.nq;
    F.out_stride = P.out_stride;
    F.lanes = (uint32_t)lanes;
    F.mode = mode;
    F.first_fail = &sc.status.as<StatusBlock>()->first_fail[0];
    F.check = P.l_check ? 1 : 0;
    if (std::getenv("NDI_TRACE_PLAN"))
      std::fprintf(stderr, "[ndi plan] lanes L=%llu qpl=%d index=%s m=%u maxk=%u tb=%u grid=%u lds=%zu prepass=%d\n", (unsigned long long)lanes,
                   P.l_qpl, pyr.dlut.lut ? "dense" : "guess", pyr.dlut.m, pyr.dlut.maxk, P.f_tb, P.f_grid, P.f_lds, P.l_check ? 0 : 1);
    constexpr int VN = Wide<T>::N;
    const dim3 grid(P.f_grid), block(P.f_tb);
#define NDI_LK(KERN)                                                                      \
  do {                                                                                    \
    auto kern = KERN;                                                                     \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)FUSED_LDS_LIMIT);         \
    launch1<T>(s, PC_EVAL, grid, block, P.f_lds, kern, F);                                \
  } while (0)
#define NDI_LS(ST, QPL)                                                                   \
  do {                                                                                    \
    if (P.f_tb == 1024) NDI_LK((eval_scalar_kernel<T, ST, QPL, 1024>));                   \
    else NDI_LK((eval_scalar_kernel<T, ST, QPL, 256>));                                   \
  } while (0)
#define NDI_LL(ST, LC)                                                                    \
  do {                                                                                    \
    if (P.f_tb == 1024) NDI_LK((eval_lanes_kernel<T, ST, LC, 1024>));                     \
    else NDI_LK((eval_lanes_kernel<T, ST, LC, 256>));                                     \
  } while (0)
#define NDI_LLC(ST)                                                                       \
  do {                                                                                    \
    if (lanes == 2) NDI_LL(ST, 2); else if (lanes == 5) NDI_LL(ST, 5);                    \
    else if (lanes == 8) NDI_LL(ST, 8); else NDI_LL(ST, 0);                               \
  } while (0)
    if (lanes == 1) {
      if (strategy == NDI_CUBIC_SPLINE) { if (P.l_qpl == VN) NDI_LS(ST_CUBIC, VN); else NDI_LS(ST_CUBIC, 1); }
      else { if (P.l_qpl == VN) NDI_LS(ST_LINEAR, VN); else NDI_LS(ST_LINEAR, 1); }
    } else {
      if (strategy == NDI_CUBIC_SPLINE) NDI_LLC(ST_CUBIC); else NDI_LLC(ST_LINEAR);
    }
#undef NDI_LLC
#undef NDI_LL
#undef NDI_LS
#undef NDI_LK
  }

  // LDS footprint of eval_fused_kernel: [pyramid | lut | per-wave strips | tables]
  size_t fused_lds_bytes(bool with_lut, unsigned tb, int tables) const {
    size_t b = (pyr.lds_bytes + 15) & ~(size_t)15;
    if (with_lut) b += pyr.lut_bytes;
    const bool strip2 = strategy != NDI_CUBIC_SPLINE || tables == 2;
    b += (size_t)(tb / 64) * 64 * (sizeof(uint32_t) + (strip2 ? 2 : 1) * sizeof(T));
    if (tables) {
      const size_t rows = tables == 2 ? 2 * n : (strategy == NDI_CUBIC_SPLINE ? 3 * n - 2 : n);
      b += rows * lanes * sizeof(T);
    }
    return b;
  }

  // Query-order fused search + evaluation (short rows): decides the variant and its launch shape, and enqueues the
  // range pre-pass the kernel relies on.  Returns false when the shape is not eligible.
  bool plan_fused(hipStream_t s, Scratch& sc, Plan1& P, const ShortKnobs& K, int flags = 0) {
    constexpr int long_axes = 1;     // (axes beyond half the LDS: one large workgroup per CU)
    constexpr int global_axes = 1;   // (axes beyond LDS: the u32 bucket index in global memory)
    const uint64_t LV = P.LV;
    if (LV == 0 || 64ull * LV * LV >= (1ull << 32) || (uint64_t)n * LV >= (1ull << 32) * 1ull) return false;   // 32-bit item / vector indices
    constexpr int lut_env = 1;   // (the bucket index: A/B settled in round 3)
    if (pyr.lds_bytes > (long_axes ? LDS_STAGE_LIMIT - 8 * 1024 : LDS_STAGE_LIMIT / 2)) {
      // Axes too long for LDS: the same kernel with the knots left in global memory and searched through the u32 bucket
      // index (TLDS == 3) -- one launch, no idx[] / t[] round trip.  Batches below 4096 queries keep the two-kernel form.
      if (!global_axes || !long_axes || !lut_env || P.nq < 4096) return false;
      pyr.ensure_bucket_index32();
      P.f_lut = false;
      P.f_unr = 2;
      P.f_tlds = 3;
      P.f_tb = 256;
      const bool strip2 = strategy != NDI_CUBIC_SPLINE;
      P.f_lds = (size_t)(256 / 64) * 64 * (sizeof(uint32_t) + (strip2 ? 2 : 1) * sizeof(T));
      P.f_pack = (K.pack > 0 || (K.pack < 0 && lanes * sizeof(T) < 128)) && ensure_packed(s);
      P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + 255) / 256, (uint64_t)cu_count() * 8 * 4));
      P.kind = Plan1::FUSED;
      g_last_path.store(NDI_PATH_GATHER);
      P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;   // fresh output: the kernel's own range test, no pre-pass
      if (P.l_check) return true;
      StatusBlock* st = sc.status.as<StatusBlock>();
      const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + BLOCK - 1) / BLOCK, 4096));
      ProfScope ps(s, PC_LOCATE);
      hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, P.q, (const T*)nullptr, P.nq, k0, kn, k0, kn,
                         mode, &st->first_fail[0]);
      NDI_HIP(hipGetLastError());
      ps.done();
      return true;
    }
    if (lut_env && P.nq >= 4096) pyr.ensure_bucket_index();
    // the bucket index beside the knots when it costs no more than half the waves a CU could hold without it
    auto waves_at = [&](bool with_lut, unsigned tb) -> size_t {
      const size_t need = fused_lds_bytes(with_lut, tb, 0);
      return need > LDS_STAGE_LIMIT ? 0 : std::min<size_t>((160 * 1024) / need, 32 / (tb / 64)) * (tb / 64);
    };
    auto best_tb = [&](bool with_lut, unsigned& tb_out) -> size_t {
      size_t best = 0;
      for (unsigned tb : {256u, 512u, 1024u}) {
        const size_t w = waves_at(with_lut, tb);
        if (w > best) { best = w; tb_out = tb; }
      }
      return best;
    };
    unsigned tb_plain = 256, tb_lut = 256;
    const size_t w_plain = best_tb(false, tb_plain);
    const size_t w_lut = (lut_env && P.nq >= 4096 && pyr.lut_bytes != 0) ? best_tb(true, tb_lut) : 0;
    if (w_plain == 0 && w_lut == 0) return false;
    P.f_lut = w_lut != 0 && 2 * w_lut >= w_plain;
    const unsigned tb_auto = P.f_lut ? tb_lut : tb_plain;
    P.f_unr = (K.unr == 1 || K.unr == 4) ? K.unr : 2;
    // Tables in LDS when they fit beside everything else, and when the batch gives every workgroup several times the
    // table size to write (the staging pass is per workgroup).  Workgroup size: the one that keeps most waves on a
    // CU beside the tables; among equals the largest (fewest staging passes).
    // {y, k} (the spline's derivatives, kept by small builds) is two thirds of {y, a, b}: it fits where the latter
    // does not, and leaves room for more waves where both do.  NDI_FUSED_LDS = 1 / 2 pins the form.
    P.f_tlds = 0;
    P.f_tb = (K.tb == 512 || K.tb == 1024) ? (unsigned)K.tb : tb_auto;
    if (K.lds != 0) {
      size_t best_waves = 0;
      const bool yk_ok = strategy == NDI_CUBIC_SPLINE && ck.p;
      const int pinned = (K.lds == 2 && !yk_ok) ? 1 : K.lds;
      for (int form : {1, 2}) {   // equal wave counts: {y, a, b} (no per-item re-forming; 2-3 % faster where both fit)
        if (form == 2 && !yk_ok) continue;
        if (pinned > 0 && pinned <= 2 && pinned != form) continue;
        for (unsigned tb : {1024u, 512u, 256u}) {
          if (K.tb && (unsigned)K.tb != tb) continue;
          const size_t need = fused_lds_bytes(P.f_lut, tb, form);
          if (need > FUSED_LDS_LIMIT) continue;
          const size_t waves = std::min<size_t>((160 * 1024) / need, 32 / (tb / 64)) * (tb / 64);
          if (waves > best_waves) {
            best_waves = waves;
            P.f_tlds = form;
            P.f_tb = tb;
          }
        }
      }
      const size_t tab = fused_lds_bytes(false, 64, P.f_tlds) - fused_lds_bytes(false, 64, 0);
      // not for small batches (the staging pass is per workgroup), and not for Linear rows of 128 B (f32: 64 B) or more:
      // two operand reads per item from L2 beat the workgroups that fit around a 64-128 KiB table (f32 x 32: 5.6 vs 4.9
      // TB/s, f32 x 16: 5.25 vs 4.76; f64 x 8 and f32 x 8 stay in LDS: 3.75 vs 2.8, 4.0 vs 2.8)
      if (P.f_tlds && K.lds < 0 && (P.nq * lanes * sizeof(T) < 8 * (size_t)cu_count() * tab ||
                                    (strategy != NDI_CUBIC_SPLINE && lanes * sizeof(T) >= (sizeof(T) == 4 ? 64 : 128)))) {
        P.f_tlds = 0;
        P.f_tb = (K.tb == 512 || K.tb == 1024) ? (unsigned)K.tb : tb_auto;
      }
    }
    P.f_lds = fused_lds_bytes(P.f_lut, P.f_tb, P.f_tlds);
    // rows shorter than a cache line read from L2: one contiguous record per interval instead of three row pieces
    P.f_pack = !P.f_tlds && (K.pack > 0 || (K.pack < 0 && lanes * sizeof(T) < 128)) && ensure_packed(s);
    // Tables from L2, rows of 64 bytes and more, a large batch on an axis of up to 4096 intervals: the workgroup-local
    // interval order (eval_fused_sorted_kernel: the operand rows of neighbouring items coincide and come from L1 -- the
    // query-order form is bound by the L2 request rate on these).  NDI_FUSED_SORTED=0 / 1: A/B.
    {
      static const bool tune_live5 = std::getenv("NDI_TUNE_LIVE") != nullptr;
      static const int fs_once = ShortKnobs::env("NDI_FUSED_SORTED", -1);
      const int fs = tune_live5 ? ShortKnobs::env("NDI_FUSED_SORTED", -1) : fs_once;
      const bool cubic = strategy == NDI_CUBIC_SPLINE;
      const size_t lds_s = ((pyr.lds_bytes + 15) & ~(size_t)15) + (P.f_lut ? pyr.lut_bytes : 0) + (((size_t)(n - 1) * 4 + 15) & ~(size_t)15) +
                           (size_t)4096 / (cubic ? 1 : 2) * (4 + (cubic ? 1 : 2) * sizeof(T));
      // AUTO: where it was measured ahead (profiles/r05_tuning.md 10): CubicSpline (four operand rows per output row), rows of
      // 128 bytes and more, at least two queries per interval and round (4096 queries: axes of up to 2049 knots)
      P.f_sorted = fs != 0 && !P.f_tlds && n >= 3 && n - 1 <= 4096 && LV >= 2 && lanes * sizeof(T) >= 64 && lds_s <= FUSED_LDS_LIMIT - 256 &&
                   (fs > 0 || (cubic && lanes * sizeof(T) >= 128 && n - 1 <= 2048 && P.nq >= (1u << 18)));
      if (P.f_sorted) {
        P.f_pack = false;
        P.f_tb = 512;
        P.f_lds = lds_s;
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds_s, 4));
        const uint64_t nqw = cubic ? 4096 : 2048;
        P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + nqw - 1) / nqw, (uint64_t)cu_count() * per_cu * 4));
        P.kind = Plan1::FUSED;
        g_last_path.store(NDI_PATH_GATHER);
        P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;
        if (P.l_check) return true;
        StatusBlock* st = sc.status.as<StatusBlock>();
        const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
        const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + BLOCK - 1) / BLOCK, 4096));
        ProfScope ps(s, PC_LOCATE);
        hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, P.q, (const T*)nullptr, P.nq, k0, kn, k0, kn,
                           mode, &st->first_fail[0]);
        NDI_HIP(hipGetLastError());
        ps.done();
        return true;
      }
    }
    const size_t wg_per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / std::max<size_t>(P.f_lds, 1), 32 / (P.f_tb / 64)));
    const uint64_t want = (uint64_t)cu_count() * (K.wgs > 0 ? (size_t)K.wgs : wg_per_cu * (P.f_tlds ? 1 : 4));
    P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + P.f_tb - 1) / P.f_tb, want));
    P.kind = Plan1::FUSED;
    g_last_path.store(NDI_PATH_GATHER);
    P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;     // fresh output: the kernel's own range test, no pre-pass
    if (P.l_check) return true;
    StatusBlock* st = sc.status.as<StatusBlock>();
    const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((P.nq + BLOCK - 1) / BLOCK, 4096));
    ProfScope ps(s, PC_LOCATE);
    hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, P.q, (const T*)nullptr, P.nq, k0, kn, k0, kn,
                       mode, &st->first_fail[0]);
    NDI_HIP(hipGetLastError());
    ps.done();
    return true;
  }

  Plan1 prep(hipStream_t s, Scratch& sc, const T* q, uint64_t nq, T* out, uint64_t out_stride, int path,
             bool beside_eval = false, int flags = 0) {
    Plan1 P;
    P.q = q; P.nq = nq; P.out = out; P.out_stride = out_stride;
    sc.status.reserve(sizeof(StatusBlock));
    reset_status(sc.status.p, s);
    StatusBlock* st = sc.status.as<StatusBlock>();

    // 1-2 lanes: one thread per query (search + evaluation in one launch, tables gathered from L2) is the latency path
    // and the faster one up to a few million queries; beyond, the query-order kernel with the tables in LDS takes over
    // (scalar data at 1e8 queries: 98 -> 157 Gqueries/s f64, 2 lanes 45 -> 141).  Measured crossover
    // (profiles/r04_scalar_crossover.jsonl): ~8e6 output elements on <= 1024 knots, proportionally earlier on longer
    // axes (8192 knots: ~1e6), whose table gathers miss L1.
    if (plan_lanes(s, sc, P, path, flags)) return P;
    constexpr long small_maxq = 8000000;
    const bool small_first = lanes <= 2 && pyr.lds_bytes <= LDS_STAGE_LIMIT;
    const double small_work = (double)nq * (double)lanes * (double)std::max<uint64_t>(n, 1024) / 1024.0;
    if (small_first && small_work >= (double)small_maxq) {
      const ShortKnobs K0 = short_knobs();
      constexpr int VN0 = Wide<T>::N;
      P.vec_ok = (lanes % VN0 == 0) && (out_stride % VN0 == 0) && aligned16(out);
      P.LV = P.vec_ok ? lanes / VN0 : lanes;
      if (K0.mode != 1 && K0.mode != 3 && plan_fused(s, sc, P, K0, flags)) return P;
    }
    if (small_first) {
      // short trailing axes: range pre-check (so rows after the first failing query stay untouched in the
      // caller's buffer), then search + evaluation fused in one launch -- no index / t round trip through HBM
      P.kind = Plan1::SMALL;
      g_last_path.store(NDI_PATH_GATHER);
      const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
      ProfScope ps(s, PC_LOCATE);
      hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, q, (const T*)nullptr, nq, k0, kn, k0, kn,
                         mode, &st->first_fail[0]);
      NDI_HIP(hipGetLastError());
      ps.done();
      return P;
    }
    constexpr int VN = Wide<T>::N;
    P.vec_ok = (lanes % VN == 0) && (out_stride % VN == 0) && aligned16(out);
    P.LV = P.vec_ok ? lanes / VN : lanes;
    const bool rows_ok = P.vec_ok && P.LV >= (uint64_t)BLOCK;
    const ShortKnobs K = short_knobs();
    // rows shorter than one workgroup pass (or unaligned): grouped from a few hundred bytes per row upwards when the
    // batch has enough queries per interval, else query order with the search fused in
    const bool short_rows = !rows_ok || (P.vec_ok && P.LV < (uint64_t)K.maxlv);
    // (axes whose interval histogram does not fit LDS are grouped with global atomics, as long rows are)
    const bool group_ok = nq < 0xffffffffull && P.vec_ok && P.LV < (uint64_t)BLOCK;
    bool bucketed = false, grouped_short = false;
    if (path == NDI_PATH_BUCKETED) {
      bucketed = rows_ok && nq < 0xffffffffull;
      grouped_short = short_rows && group_ok && K.mode != 1 && K.mode != 2;
    } else if (path == NDI_PATH_AUTO) {
      bucketed = rows_ok && nq < 0xffffffffull && nq >= 5 * (n - 1);  // measured crossover (tools/auto_threshold.py)
      // (Linear reads two operand rows per item instead of four: its query-order form already runs at 5-6 TB/s on rows
      //  of 128 B - 2 KiB and beats the grouped form everywhere: profiles/r04_short_rows_linear_sweep.txt)
      grouped_short = short_rows && group_ok && nq >= 5 * (n - 1) &&
                      (K.mode == 3 ||
                       // Linear: only when its one table outgrows L2 and the rows are 1 KiB or more (16 384 knots x 512 f32
                       // lanes, 33.6 MB: 3.54 vs 2.55 TB/s)
                       (K.mode == 0 && strategy != NDI_CUBIC_SPLINE && lanes * sizeof(T) >= 1024 &&
                        (size_t)n * lanes * sizeof(T) >= ((size_t)8 << 20)) ||
                       (K.mode == 0 && strategy == NDI_CUBIC_SPLINE &&
                                       (lanes * sizeof(T) >= (uint64_t)K.rowb ||
                                        // tables that outgrow L2 (its 4 MiB per XCD): the query-order form re-reads them
                                        // from memory per query, the grouped form once per interval -- from 512-byte rows
                                        // (8192 knots x 128 f32 lanes, 12.6 MB: 3.94 vs 2.21 TB/s; tools/group_vs_fused_probe.py)
                                        (lanes * sizeof(T) >= 512 && (size_t)(3 * n - 2) * lanes * sizeof(T) >= ((size_t)8 << 20)))));
    }
    if (short_rows && !grouped_short && !bucketed && K.mode != 1 && K.mode != 3 && plan_fused(s, sc, P, K, flags)) return P;
    g_last_path.store(bucketed || grouped_short ? NDI_PATH_BUCKETED : NDI_PATH_GATHER);
    sc.idx.reserve(nq * sizeof(uint32_t));   // the two-kernel forms: interval index (and t) per query
    if (strategy == NDI_CUBIC_SPLINE) sc.t.reserve(nq * sizeof(T));
    T* t_out = strategy == NDI_CUBIC_SPLINE ? sc.t.as<T>() : nullptr;

    if (bucketed || grouped_short) {
      P.kind = bucketed ? Plan1::BUCKETED : Plan1::BUCKETED_SHORT;
      P.s_cq = K.cq;
      const uint32_t nb = (uint32_t)(n - 1);
      sc.counts.reserve(((size_t)nb + 4) * sizeof(uint32_t));   // (+4: the fused scan reads / writes 16-byte pieces)
      sc.cursor.reserve(((size_t)nb + 4) * sizeof(uint32_t));
      sc.perm.reserve(nq * sizeof(uint4));
      const T* sval = strategy == NDI_CUBIC_SPLINE ? (const T*)sc.t.as<T>() : q;   // t (cubic) / raw x (linear)
      if (lds_sort_fits(pyr, nb)) {
        // block-local counting sort: histogram per query slice in LDS, no global atomics
        sc.hist.reserve((size_t)GROUP_MAX_BLOCKS * nb * sizeof(uint32_t));
        uint64_t slice = 0;
        uint32_t blocks = 0;
        run_locate<T>(s, pyr, q, nq, sc.idx.as<uint32_t>(), nullptr, t_out,
                      &st->first_fail[0], mode, sc.hist.as<uint32_t>(), nb, &slice, &blocks, beside_eval);
        ProfScope ps(s, PC_GROUP);
        hipLaunchKernelGGL(group_offsets_scan_kernel, dim3((nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s,
                           sc.hist.as<uint32_t>(), blocks, nb, sc.counts.as<uint32_t>(), sc.cursor.as<uint32_t>(), st,
                           nq, 0u, (uint32_t*)nullptr);
        allow_dynamic_lds(reinterpret_cast<const void*>(&group_scatter_kernel<T>), (int)(GROUP_MAX_BINS * 4));
        hipLaunchKernelGGL(group_scatter_kernel<T>, dim3(blocks), dim3(BLOCK), (size_t)nb * 4, s,
                           (const uint32_t*)sc.idx.as<uint32_t>(), sval, nq, slice,
                           (const uint32_t*)sc.hist.as<uint32_t>(), (const uint32_t*)sc.cursor.as<uint32_t>(),
                           nb, sc.perm.as<uint4>());
        NDI_HIP(hipGetLastError());
        ps.done();
      } else {
        // many intervals: histogram and placement with global atomics
        run_locate<T>(s, pyr, q, nq, sc.idx.as<uint32_t>(), nullptr, t_out, &st->first_fail[0], mode);
        ProfScope ps(s, PC_GROUP);
        NDI_HIP(hipMemsetAsync(sc.counts.p, 0, (size_t)nb * sizeof(uint32_t), s));
        const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 2048));
        hipLaunchKernelGGL(bucket_count_kernel, dim3(g), dim3(BLOCK), 0, s, (const uint32_t*)sc.idx.as<uint32_t>(), nq,
                           (const StatusBlock*)st, sc.counts.as<uint32_t>());
        hipLaunchKernelGGL(bucket_scan_kernel<256>, dim3(1), dim3(256), 0, s, sc.counts.as<uint32_t>(), nb,
                           sc.cursor.as<uint32_t>(), st);
        hipLaunchKernelGGL(bucket_scatter_kernel<T>, dim3(g), dim3(BLOCK), 0, s, (const uint32_t*)sc.idx.as<uint32_t>(),
                           sval, nq, (const StatusBlock*)st, sc.cursor.as<uint32_t>(), sc.perm.as<uint4>());
        NDI_HIP(hipGetLastError());
        ps.done();
      }
      return P;
    }
    run_locate<T>(s, pyr, q, nq, sc.idx.as<uint32_t>(), nullptr, t_out, &st->first_fail[0], mode);
    P.kind = rows_ok ? Plan1::ROWS : Plan1::FLAT;
    return P;
  }

  void launch_eval(hipStream_t s, Scratch& sc, const Plan1& P) {
    StatusBlock* st = sc.status.as<StatusBlock>();
    const uint64_t nq = P.nq;
    if (P.kind == Plan1::SMALL) {
      allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_CUBIC>), (int)LDS_STAGE_LIMIT);
      allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_LINEAR>), (int)LDS_STAGE_LIMIT);
      EvalSmallArgs<T> S{};
      S.pyr = pyr.view;
      S.data = data.as<T>();
      S.ca = ca.as<T>();
      S.cb = cb.as<T>();
      S.q = P.q;
      S.out = P.out;
      S.nq = nq;
      S.out_stride = P.out_stride;
      S.lanes = (uint32_t)lanes;
      S.mode = mode;
      S.first_fail = &st->first_fail[0];
      S.prechecked = 1;
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
      const size_t shmem = (pyr.lds_bytes + 15) & ~(size_t)15;
      if (strategy == NDI_CUBIC_SPLINE) launch1<T>(s, PC_EVAL, dim3(g), dim3(BLOCK), shmem, eval_small_kernel<T, ST_CUBIC>, S);
      else launch1<T>(s, PC_EVAL, dim3(g), dim3(BLOCK), shmem, eval_small_kernel<T, ST_LINEAR>, S);
      return;
    }
    if (P.kind == Plan1::FUSED) {
      launch_fused(s, sc, P);
      return;
    }
    if (P.kind == Plan1::LANES) {
      launch_lanes(s, sc, P);
      return;
    }
    Eval1Args<T> A{};
    A.knots = pyr.view.lv0;
    A.data = data.as<T>();
    A.ca = ca.as<T>();
    A.cb = cb.as<T>();
    A.q = P.q;
    A.idx = sc.idx.as<uint32_t>();
    A.t = sc.t.as<T>();
    A.out = P.out;
    A.lanes = lanes;
    A.out_stride = P.out_stride;
    A.nq = nq;
    A.status = st;
    A.n_int = (uint32_t)(n - 1);
#ifdef NDI_BOUNDS
    if (ShortKnobs::env("NDI_BOUNDS_SELFTEST", 0)) A.n_int = 1;   // checked build: provoke the checker (rows are then wrong)
#endif
    constexpr int VN = Wide<T>::N;
    const uint64_t LV = P.LV;
    if (P.kind == Plan1::BUCKETED) {
      A.rec = sc.perm.as<uint4>();
      constexpr int CQ = 128;
      const int U = LV >= 2048 ? 8 : (LV >= 1024 ? 4 : (LV >= 512 ? 2 : 1));
      const uint64_t segs = (LV + (uint64_t)BLOCK * U - 1) / ((uint64_t)BLOCK * U);
      // a multiple of 8 so that a workgroup keeps its XCD residue when it strides (XCD-aware chunk order);
      // every workgroup takes a run of consecutive chunks (operand rows stay in registers across them)
      const uint64_t per_xcd = ((nq + CQ - 1) / CQ + 7) / 8;
      A.run = BUCKETED_RUN;
      const uint64_t runs_per_xcd = (per_xcd + A.run - 1) / A.run;
      const unsigned gx = (unsigned)std::max<uint64_t>(8, std::min<uint64_t>(runs_per_xcd * 8, 65528));
      dim3 grid(gx, (unsigned)std::min<uint64_t>(segs, 64));
      const bool full = LV % ((uint64_t)BLOCK * U) == 0;   // whole segments only: straight-line kernel variant
#define NDI_BK(ST, UU)                                                                                      \
  do {                                                                                                      \
    if (full) launch1<T>(s, PC_EVAL, grid, dim3(BLOCK), 0, eval_bucketed_kernel<T, ST, UU, CQ, true, true>, A); \
    else launch1<T>(s, PC_EVAL, grid, dim3(BLOCK), 0, eval_bucketed_kernel<T, ST, UU, CQ, true, false>, A);   \
  } while (0)
      if (strategy == NDI_CUBIC_SPLINE) {
        if (U == 8) NDI_BK(ST_CUBIC, 8); else if (U == 4) NDI_BK(ST_CUBIC, 4);
        else if (U == 2) NDI_BK(ST_CUBIC, 2); else NDI_BK(ST_CUBIC, 1);
      } else {
        if (U == 8) NDI_BK(ST_LINEAR, 8); else if (U == 4) NDI_BK(ST_LINEAR, 4);
        else if (U == 2) NDI_BK(ST_LINEAR, 2); else NDI_BK(ST_LINEAR, 1);
      }
#undef NDI_BK
      return;
    }
    if (P.kind == Plan1::BUCKETED_SHORT) {
      A.rec = sc.perm.as<uint4>();
      uint32_t G = 8;
      while (G < LV) G *= 2;   // threads per row: power of two >= vectors per row (LV < 256)
      const int CQ = P.s_cq == 16 ? 16 : 64;
      const uint64_t wq = (uint64_t)(BLOCK / G) * CQ;
      const uint64_t per_xcd = ((nq + wq - 1) / wq + 7) / 8;
      const unsigned gx = (unsigned)std::max<uint64_t>(8, std::min<uint64_t>(per_xcd * 8, 65528));
#define NDI_BS(ST, GG)                                                                                          \
  do {                                                                                                          \
    if (CQ == 16) launch1<T>(s, PC_EVAL, dim3(gx), dim3(BLOCK), 0, eval_bucketed_short_kernel<T, ST, GG, 16>, A); \
    else launch1<T>(s, PC_EVAL, dim3(gx), dim3(BLOCK), 0, eval_bucketed_short_kernel<T, ST, GG, 64>, A);          \
  } while (0)
#define NDI_BSG(ST)                                                                   \
  do {                                                                                \
    if (G == 8) NDI_BS(ST, 8); else if (G == 16) NDI_BS(ST, 16);                      \
    else if (G == 32) NDI_BS(ST, 32); else if (G == 64) NDI_BS(ST, 64);               \
    else if (G == 128) NDI_BS(ST, 128); else NDI_BS(ST, 256);                         \
  } while (0)
      if (strategy == NDI_CUBIC_SPLINE) NDI_BSG(ST_CUBIC); else NDI_BSG(ST_LINEAR);
#undef NDI_BSG
#undef NDI_BS
      return;
    }
    if (P.kind == Plan1::ROWS) {
      // one 256-vector segment per workgroup pass and many workgroups: measured best on MI355X
      const uint64_t segs = (LV + BLOCK - 1) / BLOCK;
      const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nq, 65536));
      dim3 grid(gx, (unsigned)std::min<uint64_t>(segs, 64));
      if (strategy == NDI_CUBIC_SPLINE) launch1<T>(s, PC_EVAL, grid, dim3(BLOCK), 0, eval_rows_kernel<T, ST_CUBIC, 1>, A);
      else launch1<T>(s, PC_EVAL, grid, dim3(BLOCK), 0, eval_rows_kernel<T, ST_LINEAR, 1>, A);
      return;
    }
    // flat
    const uint32_t tile_q = (uint32_t)std::max<uint64_t>(1, 1024 / std::max<uint64_t>(LV, 1));
    const uint64_t ntiles = (nq + tile_q - 1) / tile_q;
    const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 16384));
    ProfScope ps(s, PC_EVAL);
    if (strategy == NDI_CUBIC_SPLINE) {
      if (P.vec_ok) hipLaunchKernelGGL((eval_flat_kernel<T, ST_CUBIC, VN>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
      else hipLaunchKernelGGL((eval_flat_kernel<T, ST_CUBIC, 1>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
    } else {
      if (P.vec_ok) hipLaunchKernelGGL((eval_flat_kernel<T, ST_LINEAR, VN>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
      else hipLaunchKernelGGL((eval_flat_kernel<T, ST_LINEAR, 1>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
    }
    NDI_HIP(hipGetLastError());
    ps.done();
  }

  void launch_fused(hipStream_t s, Scratch& sc, const Plan1& P) {
    EvalFusedArgs<T> F{};
    F.pyr = pyr.view;
    F.bx = P.f_lut ? pyr.bidx : BucketIndex<T>{nullptr, 0, T(0)};
    F.data = data.as<T>();
    F.ca = ca.as<T>();
    F.cb = cb.as<T>();
    F.ck = ck.as<T>();
    F.bx32 = P.f_tlds == 3 ? pyr.bidx32 : BucketIndex32<T>{nullptr, 0, T(0)};
    F.rec_stride = (uint32_t)P.LV;
    if (P.f_pack) {   // {y[i], y[i+1], a[i], b[i]} per interval
      F.data = packed.as<T>();
      F.ca = packed.as<T>() + 2 * lanes;
      F.cb = packed.as<T>() + 3 * lanes;
      F.rec_stride = (uint32_t)P.LV * (strategy == NDI_CUBIC_SPLINE ? 4u : 2u);
    }
    F.q = P.q;
    F.out = P.out;
    F.nq = P.nq;
    F.out_stride = P.out_stride;
    F.lanes = (uint32_t)lanes;
    F.lv = (uint32_t)P.LV;
    F.lv_magic = F.lv >= 2 ? (uint32_t)(((1ull << 32) + F.lv - 1) / F.lv) : 0u;
    F.mode = mode;
    F.first_fail = &sc.status.as<StatusBlock>()->first_fail[0];
    F.check = P.l_check ? 1 : 0;
    F.debug = 0;
#ifdef NDI_TUNING
    F.debug = ShortKnobs::env("NDI_FUSED_DEBUG", 0);
#endif
    constexpr int VN = Wide<T>::N;
    const dim3 grid(P.f_grid), block(P.f_tb);
    if (P.f_sorted) {
      if (std::getenv("NDI_TRACE_PLAN"))
        std::fprintf(stderr, "[ndi plan] fused sorted lut=%d tb=%u grid=%u lds=%zu prepass=%d\n", (int)P.f_lut, P.f_tb, P.f_grid,
                     P.f_lds, P.l_check ? 0 : 1);
#define NDI_FSK(ST, VEC, QPT)                                                                          \
  do {                                                                                                 \
    auto kern = eval_fused_sorted_kernel<T, ST, VEC, 512, QPT>;                                        \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(FUSED_LDS_LIMIT - 256));              \
    launch1<T>(s, PC_EVAL, grid, dim3(512), P.f_lds, kern, F);                                         \
  } while (0)
      if (strategy == NDI_CUBIC_SPLINE) { if (P.vec_ok) NDI_FSK(ST_CUBIC, VN, 8); else NDI_FSK(ST_CUBIC, 1, 8); }
      else { if (P.vec_ok) NDI_FSK(ST_LINEAR, VN, 4); else NDI_FSK(ST_LINEAR, 1, 4); }
#undef NDI_FSK
      return;
    }
    if (std::getenv("NDI_TRACE_PLAN"))   // which variant a batch took (tests assert on it; read per call)
      std::fprintf(stderr, "[ndi plan] fused tables=%s lut=%d pack=%d unr=%d tb=%u grid=%u lds=%zu prepass=%d\n",
                   P.f_tlds == 2 ? "lds{y,k}" : (P.f_tlds == 1 ? "lds{y,a,b}" : (P.f_tlds == 3 ? "memory,knots=global" : "memory")),
                   (int)P.f_lut, (int)P.f_pack,
                   P.f_unr, P.f_tb, P.f_grid, P.f_lds, P.l_check ? 0 : 1);
#define NDI_FU(ST, VEC, UNR, TB, TL)                                                                   \
  do {                                                                                                 \
    auto kern = eval_fused_kernel<T, ST, VEC, UNR, TB, TL>;                                            \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)FUSED_LDS_LIMIT);                      \
    launch1<T>(s, PC_EVAL, grid, block, P.f_lds, kern, F);                                             \
  } while (0)
#define NDI_FU_TB(ST, VEC, UNR, TL)                                      \
  do {                                                                   \
    if (P.f_tb == 1024) NDI_FU(ST, VEC, UNR, 1024, TL);                  \
    else if (P.f_tb == 512) NDI_FU(ST, VEC, UNR, 512, TL);               \
    else NDI_FU(ST, VEC, UNR, 256, TL);                                  \
  } while (0)
#define NDI_FU_UNR(ST, VEC, TL)                                          \
  do {                                                                   \
    if (P.f_unr == 4) NDI_FU_TB(ST, VEC, 4, TL);                         \
    else if (P.f_unr == 1) NDI_FU_TB(ST, VEC, 1, TL);                    \
    else NDI_FU_TB(ST, VEC, 2, TL);                                      \
  } while (0)
#define NDI_FU_ST(VEC, TL)                                                              \
  do {                                                                                  \
    if (strategy == NDI_CUBIC_SPLINE) NDI_FU_UNR(ST_CUBIC, VEC, TL);                    \
    else NDI_FU_UNR(ST_LINEAR, VEC, TL);                                                \
  } while (0)
    if (P.f_tlds == 3) {   // knots in global memory: one launch shape
      if (strategy == NDI_CUBIC_SPLINE) { if (P.vec_ok) NDI_FU(ST_CUBIC, VN, 2, 256, 3); else NDI_FU(ST_CUBIC, 1, 2, 256, 3); }
      else { if (P.vec_ok) NDI_FU(ST_LINEAR, VN, 2, 256, 3); else NDI_FU(ST_LINEAR, 1, 2, 256, 3); }
    } else if (P.vec_ok) {
      if (P.f_tlds == 2) NDI_FU_UNR(ST_CUBIC, VN, 2);
      else if (P.f_tlds) NDI_FU_ST(VN, 1);
      else NDI_FU_ST(VN, 0);
    } else {
      if (P.f_tlds == 2) NDI_FU_UNR(ST_CUBIC, 1, 2);
      else if (P.f_tlds) NDI_FU_ST(1, 1);
      else NDI_FU_ST(1, 0);
    }
#undef NDI_FU_ST
#undef NDI_FU_UNR
#undef NDI_FU_TB
#undef NDI_FU
  }

  void enqueue(hipStream_t s, Workspace& ws, const T* q, uint64_t nq, T* out, uint64_t out_stride, int path, int flags = 0) {
    launch_eval(s, ws.sc[0], prep(s, ws.sc[0], q, nq, out, out_stride, path, false, flags));
  }

  // Reads the status block of the batch enqueued with scratch set 0 (stream must be idle afterwards) and converts
  // it to the reference's error.
  ndi_status collect(hipStream_t s, Workspace& ws, uint64_t index_offset, ndi_oob_info* info) {
    ws.ensure_status();
    NDI_HIP(hipMemcpyAsync(ws.host_status, ws.sc[0].status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
    NDI_HIP(hipStreamSynchronize(s));
    ws.pending = false;
    const unsigned long long ff = ws.host_status->first_fail[0];
    if (ff == NO_FAIL) return NDI_OK;
    return report(ws.last_q, ws.last_q_space, ff, index_offset, info);
  }

  // The reference's error for the batch whose lowest failing query is q[ff] (reported as index_offset + ff).
  ndi_status report(const void* q, int q_space, unsigned long long ff, uint64_t index_offset, ndi_oob_info* info) {
    T v;
    if (q_space == NDI_MEM_DEVICE)
      NDI_HIP(hipMemcpy(&v, (const T*)q + ff, sizeof(T), hipMemcpyDeviceToHost));
    else
      v = ((const T*)q)[ff];
    // without extrapolation every failure is a range failure (NaN included: "x = NaN is not in range");
    // with it the only failure is the search meeting a NaN -- the query itself or an infinite query that the
    // periodic wrap turned into NaN (the reference panics: vector_extensions.rs:83-84)
    const ndi_status st = (mode != EX_NO) ? NDI_NAN_QUERY : NDI_OUT_OF_BOUNDS;
    if (info) {
      info->index = index_offset + ff;
      info->value = (double)v;
      info->axis = 0;
      info->status = st;
    }
    if (st == NDI_NAN_QUERY) return fail(st, "failed to convert NaN to usize (query %llu)", index_offset + ff);
    return fail(st, "x = %.17g is not in range", (double)v);
  }

  // Host queries are uploaded once per call into the workspace.
  const T* stage_queries(hipStream_t s, Workspace& ws, const void* q_, uint64_t nq, int q_space) {
    if (q_space != NDI_MEM_HOST) return (const T*)q_;
    ws.qdev.reserve(nq * sizeof(T));
    NDI_HIP(hipMemcpyAsync(ws.qdev.p, q_, nq * sizeof(T), hipMemcpyHostToDevice, s));
    return ws.qdev.as<T>();
  }

  // Range pre-pass over a whole batch (8 B per query): the lowest failing index lands in ws.host_status once the
  // stream has been synchronised.  The ring and the sharded evaluations need it before any row is produced.
  void enqueue_prepass(hipStream_t s, Workspace& ws, const T* q, uint64_t nq) {
    ws.ensure_status();
    reset_status(ws.status.p, s);
    StatusBlock* st = ws.status.as<StatusBlock>();
    const T k0 = pyr.host_knots.front(), kn = pyr.host_knots.back();
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
    {
      ProfScope ps(s, PC_LOCATE);
      hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, q, (const T*)nullptr, nq, k0, kn, k0, kn,
                         mode, &st->first_fail[0]);
      NDI_HIP(hipGetLastError());
      ps.done();
    }
    NDI_HIP(hipMemcpyAsync(ws.host_status, ws.status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
  }

  // Host output with short trailing axes (the reference's own bench shapes: scalar data, a few lanes):
  // one fused search+evaluate launch per chunk into a staging buffer the library owns, results and status
  // brought back with one synchronisation (small chunks bounce through pinned memory), and only the rows
  // before the first failing query are copied into the caller's buffer.
  ndi_status eval_small_host(hipStream_t s, Workspace& ws, const T* q_dev, const T* q_orig, int q_space,
                             uint64_t nq, T* out, uint64_t out_stride, ndi_oob_info* info) {
    const uint64_t row_bytes = lanes * sizeof(T);
    const uint64_t chunk_q = std::max<uint64_t>(1, std::min<uint64_t>(nq, (64ull << 20) / row_bytes));
    constexpr size_t BOUNCE = 8ull << 20;
    ws.stage.reserve(chunk_q * row_bytes);
    ws.ensure_status();
    g_last_path.store(NDI_PATH_GATHER);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_CUBIC>), (int)LDS_STAGE_LIMIT);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_LINEAR>), (int)LDS_STAGE_LIMIT);
    StatusBlock* st = ws.sc[0].status.as<StatusBlock>();
    for (uint64_t off = 0; off < nq; off += chunk_q) {
      const uint64_t cq = std::min<uint64_t>(chunk_q, nq - off);
      const size_t bytes = cq * row_bytes;
      NDI_HIP(hipMemsetAsync(st, 0xFF, 2 * sizeof(unsigned long long), s));
      EvalSmallArgs<T> A{};
      A.pyr = pyr.view;
      A.data = data.as<T>();
      A.ca = ca.as<T>();
      A.cb = cb.as<T>();
      A.q = q_dev + off;
      A.out = ws.stage.as<T>();
      A.nq = cq;
      A.out_stride = lanes;
      A.lanes = (uint32_t)lanes;
      A.mode = mode;
      A.first_fail = &st->first_fail[0];
      A.prechecked = 0;
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((cq + BLOCK - 1) / BLOCK, 4096));
      const size_t shmem = (pyr.lds_bytes + 15) & ~(size_t)15;
      if (strategy == NDI_CUBIC_SPLINE) launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small_kernel<T, ST_CUBIC>, A);
      else launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small_kernel<T, ST_LINEAR>, A);
      const bool bounce = bytes <= BOUNCE;
      if (bounce) {
        ws.ensure_pin(BOUNCE);
        NDI_HIP(hipMemcpyAsync(ws.pin, ws.stage.p, bytes, hipMemcpyDeviceToHost, s));
      }
      NDI_HIP(hipMemcpyAsync(ws.host_status, st, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
      NDI_HIP(hipStreamSynchronize(s));
      const unsigned long long ff = ws.host_status->first_fail[0];
      const uint64_t good = (ff == NO_FAIL) ? cq : (uint64_t)ff;
      T* dst = out + off * out_stride;
      if (good) {
        if (bounce) {
          if (out_stride == lanes) std::memcpy(dst, ws.pin, good * row_bytes);
          else
            for (uint64_t r = 0; r < good; ++r)
              std::memcpy(dst + r * out_stride, (const char*)ws.pin + r * row_bytes, row_bytes);
        } else {
          NDI_HIP(hipMemcpy2D(dst, out_stride * sizeof(T), ws.stage.p, row_bytes, row_bytes, good,
                              hipMemcpyDeviceToHost));
        }
      }
      if (ff != NO_FAIL) return report(q_orig + off, q_space, ff, off, info);
    }
    return NDI_OK;
  }

  // Host arrays in and out, short trailing axes, small batch (the reference's own bench shapes: 1e4 queries on scalar
  // data): ZERO-COPY.  The queries are copied into the workspace's pinned buffer with a plain memcpy, the fused
  // search + evaluation kernel reads them and writes the rows straight through the host mapping of that buffer (a
  // hipHostMalloc allocation is device-accessible), and one synchronisation later the rows are memcpy'd to the
  // caller -- no H2D / D2H copy commands at all, only the 32-byte status read-back.  Saves two DMA round trips per
  // call (C1: 48.7 -> see DESIGN.md 4.2).  Rows at / after the first failing query are not copied out.
  static constexpr size_t ZERO_COPY_LIMIT = 1u << 20;   // queries + rows
  bool zero_copy_fits(uint64_t nq, int q_space, int out_space) const {
    return q_space == NDI_MEM_HOST && out_space == NDI_MEM_HOST && lanes <= (uint64_t)SMALL_LANES &&
           pyr.lds_bytes <= LDS_STAGE_LIMIT && nq * (lanes + 1) * sizeof(T) <= ZERO_COPY_LIMIT;
  }
  ndi_status eval_small_zero_copy(hipStream_t s, Workspace& ws, const T* q_host, uint64_t nq, T* out,
                                  uint64_t out_stride, ndi_oob_info* info) {
    const size_t q_bytes = ((nq * sizeof(T)) + 255) & ~(size_t)255, row_bytes = lanes * sizeof(T);
    ws.ensure_pin(std::max<size_t>(q_bytes + nq * row_bytes, 8ull << 20));
    ws.ensure_status();
    T* pq = reinterpret_cast<T*>(ws.pin);
    T* po = reinterpret_cast<T*>((char*)ws.pin + q_bytes);
    std::memcpy(pq, q_host, nq * sizeof(T));
    g_last_path.store(NDI_PATH_GATHER);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_CUBIC>), (int)LDS_STAGE_LIMIT);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small_kernel<T, ST_LINEAR>), (int)LDS_STAGE_LIMIT);
    StatusBlock* st = ws.sc[0].status.as<StatusBlock>();
    NDI_HIP(hipMemsetAsync(st, 0xFF, 2 * sizeof(unsigned long long), s));
    EvalSmallArgs<T> A{};
    A.pyr = pyr.view;
    A.data = data.as<T>();
    A.ca = ca.as<T>();
    A.cb = cb.as<T>();
    A.q = ws.pin_device<const T>(pq);
    A.out = ws.pin_device<T>(po);
    A.nq = nq;
    A.out_stride = lanes;
    A.lanes = (uint32_t)lanes;
    A.mode = mode;
    A.first_fail = &st->first_fail[0];
    A.prechecked = 0;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
    const size_t shmem = (pyr.lds_bytes + 15) & ~(size_t)15;
    if (strategy == NDI_CUBIC_SPLINE) launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small_kernel<T, ST_CUBIC>, A);
    else launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small_kernel<T, ST_LINEAR>, A);
    NDI_HIP(hipMemcpyAsync(ws.host_status, st, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
    NDI_HIP(hipStreamSynchronize(s));
    const unsigned long long ff = ws.host_status->first_fail[0];
    const uint64_t good = (ff == NO_FAIL) ? nq : (uint64_t)ff;
    if (good) {
      if (out_stride == lanes) std::memcpy(out, po, good * row_bytes);
      else
        for (uint64_t r = 0; r < good; ++r) std::memcpy(out + r * out_stride, (const char*)po + r * row_bytes, row_bytes);
    }
    if (ff != NO_FAIL) return report(q_host, NDI_MEM_HOST, ff, 0, info);
    return NDI_OK;
  }

  // interp_array_into on staged (device) queries; q_orig / q_space name the caller's array for error reports.
  ndi_status eval_body(hipStream_t s, Workspace& ws, const T* q, const void* q_orig, int q_space, uint64_t nq,
                       void* out_, uint64_t out_stride, const ndi_eval_opts& o, ndi_oob_info* info) {
    ws.last_q = q_orig;
    ws.last_q_space = q_space;
    ws.last_nq = nq;
    if (o.out_memspace == NDI_MEM_DEVICE) {
      enqueue(s, ws, q, nq, (T*)out_, out_stride, o.path, o.flags);
      ws.pending = true;
      if (o.async_launch) return NDI_OK;
      return collect(s, ws, 0, info);
    }
    // host output: stream the batch through a device staging buffer in query chunks
    const uint64_t row_bytes = lanes * sizeof(T);
    if (lanes <= (uint64_t)SMALL_LANES && pyr.lds_bytes <= LDS_STAGE_LIMIT)
      return eval_small_host(s, ws, q, (const T*)q_orig, q_space, nq, (T*)out_, out_stride, info);
    const uint64_t chunk_q = std::max<uint64_t>(1, std::min<uint64_t>(nq, (256ull << 20) / row_bytes));
    ws.stage.reserve(chunk_q * row_bytes);
    ws.ensure_status();
    for (uint64_t off = 0; off < nq; off += chunk_q) {
      const uint64_t cq = std::min<uint64_t>(chunk_q, nq - off);
      enqueue(s, ws, q + off, cq, ws.stage.as<T>(), lanes, o.path);
      NDI_HIP(hipMemcpyAsync(ws.host_status, ws.sc[0].status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
      NDI_HIP(hipStreamSynchronize(s));
      unsigned long long ff = ws.host_status->first_fail[0];
      const uint64_t good = (ff == NO_FAIL) ? cq : (uint64_t)ff;
      if (good)
        NDI_HIP(hipMemcpy2D((T*)out_ + off * out_stride, out_stride * sizeof(T), ws.stage.p, row_bytes,
                            row_bytes, good, hipMemcpyDeviceToHost));
      if (ff != NO_FAIL) return report((const T*)q_orig + off, q_space, ff, off, info);
    }
    return NDI_OK;
  }

  ndi_status eval(const void* q_, uint64_t nq, void* out_, uint64_t out_stride,
                  const ndi_eval_opts* opts, ndi_oob_info* info) override {
    DeviceGuard dg(device);
    Range rg("ndi_interp1d_eval");
    ndi_eval_opts o{};
    if (const ndi_status vs__ = take_opts(opts, o); vs__ != NDI_OK) return vs__;
    hipStream_t s = (hipStream_t)o.stream;  // NULL = the HIP default stream
    if (out_stride < lanes) return fail(NDI_BAD_ARG, "out_row_stride (%llu) < lanes (%llu)",
                                        (unsigned long long)out_stride, (unsigned long long)lanes);
    if (nq == 0) return NDI_OK;
    if (!q_ || !out_) return fail(NDI_BAD_ARG, "null query / output pointer");
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    if (zero_copy_fits(nq, o.q_memspace, o.out_memspace))
      return eval_small_zero_copy(s, ws, (const T*)q_, nq, (T*)out_, out_stride, info);
    const T* q = stage_queries(s, ws, q_, nq, o.q_memspace);
    return eval_body(s, ws, q, q_, o.q_memspace, nq, out_, out_stride, o, info);
  }

  ndi_status finish(void* stream, ndi_oob_info* info) override {
    DeviceGuard dg(device);
    hipStream_t s = (hipStream_t)stream;
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    if (!ws.pending) {
      NDI_HIP(hipStreamSynchronize(s));
      return NDI_OK;
    }
    return collect(s, ws, 0, info);
  }

  // ---- ring evaluation ----------------------------------------------------------------------
  // Interp1D::interp_array for outputs that do not fit / need not stay in device memory: chunks through a ring.
  // The producer is a two-stream pipeline: locate + group of chunk k+1 run on the workspace's side stream into the
  // other scratch set while chunk k is evaluated on the caller's stream; events order the two.
  struct RingRun {
    std::unique_lock<std::mutex> own;
    std::vector<void*> slots;
    uint64_t pitch = 0, chunk = 0, cq0 = 0;
    uint32_t n_slots = 0;
    Plan1 plan0;
    hipStream_t side = nullptr;
  };

  // Resolves the ring and starts locate + group of chunk 0 on the side stream -- before the first failing index
  // of the batch is known on the host (it does not depend on it: the evaluation kernels skip rows at / after the
  // chunk's own first failure), so the range pre-pass and its synchronisation are hidden behind it.
  void ring_begin(hipStream_t s, Workspace& ws, const T* q, uint64_t nq, const ndi_ring_desc* ring, uint64_t stride,
                  const ndi_eval_opts& o, RingRun& R) {
    R.n_slots = ring->n_slots;
    R.chunk = ring->chunk_queries;
    R.slots.resize(ring->n_slots);
    R.pitch = stride;          // row pitch of a chunk, in elements
    if (ring->slots) {
      for (uint32_t i = 0; i < ring->n_slots; ++i) R.slots[i] = ring->slots[i];
    } else {
      R.own = std::unique_lock<std::mutex>(ring_own.mu);   // library-owned ring: one allocation, slots
      ring_own.ensure(ring->n_slots, ring->chunk_queries, stride * sizeof(T));   // interleaved row by row (OwnedRing)
      for (uint32_t i = 0; i < ring->n_slots; ++i) R.slots[i] = (char*)ring_own.buf.p + (size_t)i * stride * sizeof(T);
      R.pitch = (uint64_t)ring->n_slots * stride;
    }
    R.side = ring_overlap() ? ws.side_stream() : s;
    for (Scratch& sc : ws.sc) sc.ensure_events();
    // the side stream starts after everything already enqueued on s (the query upload)
    NDI_HIP(hipEventRecord(ws.order_event(), s));
    NDI_HIP(hipStreamWaitEvent(R.side, ws.order_event(), 0));
    R.cq0 = std::min<uint64_t>(R.chunk, nq);
    R.plan0 = prep(R.side, ws.sc[0], q, R.cq0, (T*)R.slots[0], R.pitch, o.path);
    NDI_HIP(hipEventRecord(ws.sc[0].prep_done, R.side));
  }

  // Produces the rows [0, limit) of the batch chunk by chunk.  q_offset / shard: position of this batch in a
  // sharded evaluation (the consumer sees global query indices).
  void ring_produce(hipStream_t s, Workspace& ws, const T* q, uint64_t limit, RingRun& R, ndi_ring_consumer consume,
                    void* user, const ndi_eval_opts& o, uint64_t q_offset, uint32_t shard) {
    std::vector<hipEvent_t> busy(R.n_slots, nullptr);
    uint64_t k = 0;
    for (uint64_t off = 0; off < limit; off += R.chunk, ++k) {
      const uint64_t cq = std::min<uint64_t>(R.chunk, limit - off);
      const uint32_t slot = (uint32_t)(k % R.n_slots);
      Scratch& sc = ws.sc[k & 1];
      Plan1 P = R.plan0;
      if (k > 0) {
        if (k >= 2) NDI_HIP(hipStreamWaitEvent(R.side, sc.eval_done, 0));   // chunk k-2 has released the set
        P = prep(R.side, sc, q + off, cq, (T*)R.slots[slot], R.pitch, o.path, R.side != s);
        NDI_HIP(hipEventRecord(sc.prep_done, R.side));
      }
      NDI_HIP(hipStreamWaitEvent(s, sc.prep_done, 0));
      if (busy[slot]) {   // the consumer reads this slot on another stream: wait for it there
        NDI_HIP(hipStreamWaitEvent(s, busy[slot], 0));
        busy[slot] = nullptr;
      }
      launch_eval(s, sc, P);
      NDI_HIP(hipEventRecord(sc.eval_done, s));
      if (consume) {
        ndi_ring_chunk c{};
        c.index = k; c.q_begin = q_offset + off; c.q_count = cq; c.out = R.slots[slot]; c.row_stride = R.pitch;
        c.slot = slot; c.shard = shard; c.stream = (void*)s;
        busy[slot] = (hipEvent_t)consume(user, &c);
      }
    }
    NDI_HIP(hipStreamSynchronize(s));
    NDI_HIP(hipStreamSynchronize(R.side));   // (a speculative chunk 0 that was never evaluated)
    for (hipEvent_t e : busy)
      if (e) NDI_HIP(hipEventSynchronize(e));
    ws.pending = false;
  }

  ndi_status eval_ring(const void* q_, uint64_t nq, const ndi_ring_desc* ring, ndi_ring_consumer consume,
                       void* user, const ndi_eval_opts* opts, ndi_oob_info* info) override {
    DeviceGuard dg(device);
    ndi_eval_opts o{};
    if (const ndi_status vs__ = take_opts(opts, o); vs__ != NDI_OK) return vs__;
    hipStream_t s = (hipStream_t)o.stream;
    uint64_t stride = 0;
    ndi_status rs = check_ring_desc(ring, lanes, &stride);
    if (rs != NDI_OK) return rs;
    if (nq == 0) return NDI_OK;
    if (!q_) return fail(NDI_BAD_ARG, "null query pointer");
    Range rg("ndi_interp1d_eval_ring");
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    const T* q = stage_queries(s, ws, q_, nq, o.q_memspace);
    // range pre-pass over the whole batch: the first failing index is known before any chunk is handed out
    enqueue_prepass(s, ws, q, nq);
    RingRun R;
    ring_begin(s, ws, q, nq, ring, stride, o, R);
    NDI_HIP(hipStreamSynchronize(s));
    const unsigned long long ff = ws.host_status->first_fail[0];
    const uint64_t limit = ff == NO_FAIL ? nq : std::min<uint64_t>(nq, ff);
    ring_produce(s, ws, q, limit, R, consume, user, o, 0, 0);
    if (ff == NO_FAIL) return NDI_OK;
    return report(q_, o.q_memspace, ff, 0, info);
  }

  ndi_status trim() override {
    DeviceGuard dg(device);
    spaces.trim();
    std::lock_guard<std::mutex> g(ring_own.mu);
    ring_own.clear();
    return NDI_OK;
  }

  uint64_t scratch_sets() override { return spaces.size(); }

  // A replica on another (or the same) device: the knot pyramid is rebuilt from the host copy, data and the spline
  // tables are copied device to device (over xGMI between two GPUs) -- no second upload of the caller's arrays, no
  // second Thomas solve.
  ndi_status clone_to(int dev, Interp1DBase** out) override {
    std::unique_ptr<Interp1DImpl<T>> h(new Interp1DImpl<T>());
    h->dtype = dtype; h->device = dev; h->lanes = lanes;
    h->strategy = strategy; h->mode = mode; h->n = n;
    {
      DeviceGuard dg(device);
      NDI_HIP(hipDeviceSynchronize());     // the source tables are complete
    }
    DeviceGuard dg(dev);
    h->pyr.upload(pyr.host_knots.data(), n);
    auto copy = [&](DevBuf& dst, const DevBuf& src) {
      if (!src.p) return;
      dst.reserve(src.bytes);
      copy_across_devices(dst.p, dev, src.p, device, src.bytes);
    };
    copy(h->data, data);
    copy(h->ca, ca);
    copy(h->cb, cb);
    copy(h->ck, ck);
    *out = h.release();
    return NDI_OK;
  }

  ndi_status coefficients(void* a_out, void* b_out, int memspace) override {
    DeviceGuard dg(device);
    if (strategy != NDI_CUBIC_SPLINE) return fail(NDI_BAD_ARG, "Linear has no coefficient tables");
    const size_t tab = (size_t)(n - 1) * lanes * sizeof(T);
    const hipMemcpyKind k = memspace == NDI_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (a_out) NDI_HIP(hipMemcpy(a_out, ca.p, tab, k));
    if (b_out) NDI_HIP(hipMemcpy(b_out, cb.p, tab, k));
    return NDI_OK;
  }
};

template <class T>
static std::vector<T> default_axis(uint64_t n) {
  std::vector<T> v(n);
  for (uint64_t i = 0; i < n; ++i) v[i] = (T)i;  // interp1d/mod.rs:402-406
  return v;
}

template <class T>
static std::vector<T> fetch_axis(const void* p, uint64_t len, int memspace) {
  std::vector<T> v(len);
  if (len == 0) return v;
  if (memspace == NDI_MEM_DEVICE)
    NDI_HIP(hipMemcpy(v.data(), p, len * sizeof(T), hipMemcpyDeviceToHost));
  else
    std::memcpy(v.data(), p, len * sizeof(T));
  return v;
}

template <class T>
static ndi_status create1d(const ndi_interp1d_desc& d, Interp1DBase** out) {
  DeviceGuard dg(d.device);
  Range rg("ndi_interp1d_create");
  std::unique_ptr<Interp1DImpl<T>> h(new Interp1DImpl<T>());
  h->dtype = d.dtype;
  h->device = d.device;
  h->strategy = d.strategy;
  h->mode = d.extrapolate ? EX_YES : EX_NO;
  h->n = d.n;
  h->lanes = d.lanes;
  BuildClock clk;
  std::vector<T> x = d.x ? fetch_axis<T>(d.x, d.x_len, d.memspace) : default_axis<T>(d.n);
  clk.mark("fetch axis");
  const uint64_t x_len = d.x ? d.x_len : d.n;
  if (d.validate) {
    ndi_status st = check_axis_1d<T>(x.data(), x_len, d.n, d.strategy);
    if (st != NDI_OK) return st;
  } else if (x_len != d.n || d.n < min_len_1d(d.strategy)) {
    return fail(NDI_BAD_ARG, "unvalidated create with inconsistent sizes (x_len %llu, n %llu)",
                (unsigned long long)x_len, (unsigned long long)d.n);
  }
  if (d.lanes == 0) return fail(NDI_BAD_ARG, "lanes must be >= 1");
  if (d.n > MAX_KNOTS) return fail(NDI_UNSUPPORTED, "more than %llu knots", (unsigned long long)MAX_KNOTS);
  if (!d.data) return fail(NDI_BAD_ARG, "null data pointer");
  clk.mark("validate");
  {   // small handles: one allocation for the knot pyramid, the data and the spline's tables (a / b / k)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint64_t block = 1;
    while ((uint64_t)64 * block < d.n) block *= 2;
    const size_t pyr_b = al((d.n + (d.n + block - 1) / block) * sizeof(T));
    const size_t data_b = al((size_t)d.n * d.lanes * sizeof(T));
    const size_t tab_b = d.strategy == NDI_CUBIC_SPLINE ? al((size_t)(d.n - 1) * d.lanes * sizeof(T)) : 0;
    const size_t k_b = (d.strategy == NDI_CUBIC_SPLINE && 2 * data_b <= FUSED_LDS_LIMIT) ? data_b : 0;
    const size_t total = pyr_b + data_b + 2 * tab_b + k_b;
    constexpr int arena_env = 1;
    if (arena_env && total <= ((size_t)1 << 20)) {
      h->arena.reserve(total);
      char* p0 = static_cast<char*>(h->arena.p);
      h->pyr.buf.adopt(p0, pyr_b);
      h->data.adopt(p0 + pyr_b, data_b);
      if (tab_b) {
        h->ca.adopt(p0 + pyr_b + data_b, tab_b);
        h->cb.adopt(p0 + pyr_b + data_b + tab_b, tab_b);
      }
      if (k_b) h->ck.adopt(p0 + pyr_b + data_b + 2 * tab_b, k_b);
    }
  }
  h->pyr.upload(x.data(), d.n);
  clk.mark("knot pyramid");
  const size_t bytes = (size_t)d.n * d.lanes * sizeof(T);
  h->data.reserve(bytes);
  NDI_HIP(hipMemcpy(h->data.p, d.data, bytes,
                    d.memspace == NDI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  clk.mark("data copy");
  if (d.strategy == NDI_CUBIC_SPLINE) {
    ndi_status st = h->build_spline(d);
    if (st != NDI_OK) return st;
    clk.mark("build_spline");
  }
  *out = h.release();
  return NDI_OK;
}

// ---------------------------------------------------------------------------------------------
// Interp2D (Bilinear)
// ---------------------------------------------------------------------------------------------
struct Interp2DBase {
  virtual ~Interp2DBase() = default;
  virtual uint64_t signature() const = 0;
  int dtype = 0, device = 0;
  uint64_t lanes = 0;
  virtual ndi_status eval(const void* qx, const void* qy, uint64_t nq, void* out, uint64_t out_stride,
                          const ndi_eval_opts* opts, ndi_oob_info* info) = 0;
  virtual ndi_status finish(void* stream, ndi_oob_info* info) = 0;
  virtual ndi_status eval_ring(const void* qx, const void* qy, uint64_t nq, const ndi_ring_desc* ring,
                               ndi_ring_consumer consume, void* user, const ndi_eval_opts* opts,
                               ndi_oob_info* info) = 0;
  virtual ndi_status trim() = 0;
  virtual ndi_status probe_ceiling(uint64_t nq, void* out, uint64_t out_stride, void* stream, int reps, double* ms) = 0;
  virtual ndi_status clone_to(int dev, Interp2DBase** out) = 0;
};

template <class T>
struct Interp2DImpl final : Interp2DBase {
  int mode = EX_NO;
  uint64_t nx = 0, ny = 0;
  DevicePyramid<T> px, py;
  DevBuf data;
  bool pair_packed = false;   // data holds the pair-packed layout (pack_pairs_kernel)
  SpaceSet spaces;
  OwnedRing ring_own;
  // lazy copies of the grid are built on first use by a kernel on the caller's stream, as Interp1DImpl::ensure_packed builds
  // its copy -- state 0 = not built, 1 = built and complete, 2 = not available, 3 = enqueued on *_stream, completion
  // signalled by *_ev
  ~Interp2DImpl() {
    if (slopes_ev) (void)hipEventDestroy(slopes_ev);
  }
  // the slope-record copy (slope_pack_kernel, eval_slopes2d_kernel): {z, m} per grid point of rows 0 .. nx - 2, 2 x the
  // grid, built on first use
  DevBuf slopes;
  std::atomic<int> slopes_state{0};
  std::mutex slopes_mu;
  hipEvent_t slopes_ev = nullptr;
  hipStream_t slopes_stream = nullptr;
  static constexpr size_t SLOPES_LIMIT = (size_t)256 << 20;
  // Layout of the copy: POINT records (2 x the grid; a query reads 4 L sizeof(T) contiguous bytes at the alignment of one
  // record) or CELL records (every cell owns its 4 L values at a stride padded to a power of two / a multiple of 128 bytes: a
  // query then touches ceil(bytes / 128) lines exactly).  The kernel is bound by L1 misses in flight (counters:
  // profiles/r06_slopes2d_*), so the layout with fewer lines per query wins; a tie goes to the smaller table.
  size_t slopes_cell_stride_bytes() const {
    const size_t need = 4 * lanes * sizeof(T);
    size_t sb = 32;
    while (sb < need && sb < 128) sb *= 2;
    if (sb < need) sb = (need + 127) / 128 * 128;
    return sb;
  }
  bool slopes_cells() const {
    const double need = (double)(4 * lanes * sizeof(T));
    const double lines_point = 1.0 + (need - 2.0 * sizeof(T)) / 128.0;
    const double lines_cell = (double)((slopes_cell_stride_bytes() + 127) / 128);
    return lines_cell <= 0.85 * lines_point;
  }
  size_t slopes_bytes() const {
    return slopes_cells() ? (size_t)(nx - 1) * (ny - 1) * slopes_cell_stride_bytes() : (size_t)(nx - 1) * ny * 2 * lanes * sizeof(T);
  }
  bool ensure_slopes(hipStream_t s) {
    int st = slopes_state.load(std::memory_order_acquire);
    if (st == 1) return true;
    if (st == 2) return false;
    std::lock_guard<std::mutex> g(slopes_mu);
    st = slopes_state.load(std::memory_order_acquire);
    if (st == 1) return true;
    if (st == 2) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
    if (cs != hipStreamCaptureStatusNone) return false;   // this batch takes another kernel
    if (st == 0) {
      const size_t bytes = slopes_bytes();
      if (bytes == 0 || bytes > SLOPES_LIMIT) {
        slopes_state.store(2, std::memory_order_release);
        return false;
      }
      try {
        maybe_fail_lazy_alloc(__LINE__);
        slopes.reserve(bytes);
        if (!slopes_ev) NDI_HIP(hipEventCreateWithFlags(&slopes_ev, hipEventDisableTiming));
        const uint64_t total = (uint64_t)(nx - 1) * ny * 2 * lanes;
        const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((total + BLOCK - 1) / BLOCK, 65536));
        if (slopes_cells()) NDI_HIP(hipMemsetAsync(slopes.p, 0, bytes, s));     // (the padding is never read; kept defined)
        hipLaunchKernelGGL(slope_pack_kernel<T>, dim3(gr), dim3(BLOCK), 0, s, (const T*)data.as<T>(), (const T*)px.view.lv0,
                           slopes.as<T>(), nx, ny, lanes, (uint64_t)(pair_packed ? ny - 1 : ny),
                           (uint64_t)(pair_packed ? 2 * lanes : lanes),
                           (uint64_t)(slopes_cells() ? slopes_cell_stride_bytes() / sizeof(T) : 0));
        NDI_HIP(hipGetLastError());
        NDI_HIP(hipEventRecord(slopes_ev, s));
      } catch (const HipFailure&) {
        (void)hipGetLastError();
        slopes.release();
        slopes_state.store(2, std::memory_order_release);
        return false;
      }
      slopes_stream = s;
      slopes_state.store(3, std::memory_order_release);
      return true;
    }
    if (hipEventQuery(slopes_ev) == hipSuccess) {   // st == 3
      slopes_state.store(1, std::memory_order_release);
      return true;
    }
    (void)hipGetLastError();
    if (s != slopes_stream) NDI_HIP(hipStreamWaitEvent(s, slopes_ev, 0));
    return true;
  }
  uint64_t signature() const override {
    uint64_t h = fnv1a(FNV_SEED, px.host_knots.data(), px.host_knots.size() * sizeof(T));
    h = fnv1a(h, py.host_knots.data(), py.host_knots.size() * sizeof(T));
    const uint64_t f[3] = {nx, ny, (uint64_t)mode};
    return fnv1a(h, f, sizeof(f));
  }

  // Two stages as in Interp1DImpl: prep() = both searches (+ the optional tile grouping) into a scratch set,
  // launch_eval() = the bilinear kernel reading that set.
  struct Plan2 {
    enum Kind { SMALL, GATHER, TILED, FUSED2, LANES2, SLOPES2 } kind = GATHER;
    int l_qpl = 1;          // LANES2, scalar grids: queries per lane (1, or one 16-byte vector)
    bool l_check = false;   // LANES2: no range pre-pass (NDI_EVAL_FRESH_OUTPUT)
    bool f_lds_wide = false;   // SLOPES2: f_lds includes the result strip of the 16-byte store path
    // FUSED2 (eval_fused2d_kernel)
    bool f_vec = false, f_lut = false;
    uint64_t f_lv = 0;
    unsigned f_grid = 1, f_tb = 256;
    size_t f_lds = 0;
    bool compact = false;   // TILED: self-contained 16-byte records (group_scatter2d_kernel<T, true>)
    uint32_t ts = 0, nty = 0, nb = 0;   // TILED: tile shift, tiles per grid row, number of tiles
    const T* qx = nullptr;
    const T* qy = nullptr;
    uint64_t nq = 0;
    T* out = nullptr;
    uint64_t out_stride = 0;
  };

  Plan2 prep(hipStream_t s, Scratch& sc, const T* qx, const T* qy, uint64_t nq, T* out, uint64_t out_stride,
             int path, bool beside_eval = false, int flags = 0) {
    Plan2 P;
    P.qx = qx; P.qy = qy; P.nq = nq; P.out = out; P.out_stride = out_stride;
    sc.idx.reserve(nq * sizeof(uint32_t));
    sc.idx2.reserve(nq * sizeof(uint32_t));
    sc.status.reserve(sizeof(StatusBlock));
    reset_status(sc.status.p, s);
    StatusBlock* st = sc.status.as<StatusBlock>();
    const size_t both = ((px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15);
    // 1-2 values per grid point: one query per thread -- both searches and the evaluation in one launch, no (xi, yi)
    // round trip, two reciprocals per query instead of three divisions per value.  (NDI_SMALL2D_LANES extends it to
    // unaligned rows of up to that many values for A/B runs: measured SLOWER from 3 values -- 100 x 100 x 5 f64: 11 vs
    // 21 Gqueries/s -- a thread per query turns every operand load into 64 scattered sectors, where the item-per-lane
    // gather kernel reads each query's 40-byte segments whole.)
    // Grids that fit LDS beside their axes (the reference's 100 x 100 scalar grid: 80 KB in f64), rows of up to 64 bytes,
    // large batches: query per lane with every corner read served by LDS (eval_scalar2d_kernel / eval_lanes2d_kernel).
    // NDI_LANES2D_KERNEL=0: A/B.
    {
      static const bool tune_live2 = std::getenv("NDI_TUNE_LIVE") != nullptr;
      static const int on_once = ShortKnobs::env("NDI_LANES2D_KERNEL", -1);
      const int on = tune_live2 ? ShortKnobs::env("NDI_LANES2D_KERNEL", -1) : on_once;
      const size_t grid_b = (size_t)nx * ny * lanes * sizeof(T);
      if (on != 0 && path != NDI_PATH_BUCKETED && !pair_packed && nx <= 16384 && ny <= 16384 && lanes * sizeof(T) <= (on > 0 ? 64 : 56) &&
          grid_b + (nx + ny) * 5 * sizeof(T) <= FUSED_LDS_LIMIT && (uint64_t)nx * ny * lanes < (1ull << 31) &&
          (on > 0 || (nq >= 65536 && (double)nq * (double)lanes * sizeof(T) >= 4.0 * (double)cu_count() * (double)grid_b))) {
        px.ensure_dense_lut();
        py.ensure_dense_lut();
        const size_t fixed = (((size_t)(nx + LANE_SENTINELS) * sizeof(T) + 15) & ~(size_t)15) +
                             (((size_t)(ny + LANE_SENTINELS) * sizeof(T) + 15) & ~(size_t)15) +
                             px.dlut_bytes + py.dlut_bytes + (size_t)(nx - 1 + ny - 1) * 4 * sizeof(T) + ((grid_b + 15) & ~(size_t)15);
        auto need_of = [&](unsigned tb) { return fixed + (lanes > 1 ? (size_t)(tb / 64) * 64 * lanes * sizeof(T) : 0); };
        unsigned tb = need_of(256) * 4 <= 160 * 1024 ? 256u : 1024u;
        if (need_of(tb) > FUSED_LDS_LIMIT) tb = 256u;   // (the strips of 16 waves do not fit: 4 waves)
        if (px.dense_ok && py.dense_ok && need_of(tb) <= FUSED_LDS_LIMIT) {
          constexpr int VNl = Wide<T>::N;
          P.kind = Plan2::LANES2;
          P.f_tb = tb;
          P.f_lds = need_of(tb);
          P.l_qpl = (lanes == 1 && out_stride == 1 && aligned16(qx) && aligned16(qy) && aligned16(out)) ? VNl : 1;
          const size_t wg_per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / P.f_lds, 32 / (P.f_tb / 64)));
          const uint64_t per_wg = (uint64_t)P.f_tb * (lanes == 1 ? (uint64_t)P.l_qpl : 1);
          P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + per_wg - 1) / per_wg, (uint64_t)cu_count() * wg_per_cu));
          g_last_path.store(NDI_PATH_GATHER);
          P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;   // fresh output: the kernel's own range test, no pre-pass
          if (P.l_check) return P;
          const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
          ProfScope ps(s, PC_LOCATE);
          hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, qx, qy, nq, px.host_knots.front(),
                             px.host_knots.back(), py.host_knots.front(), py.host_knots.back(), mode, &st->first_fail[0]);
          NDI_HIP(hipGetLastError());
          ps.done();
          return P;
        }
      }
    }
    // Short rows (up to 64 bytes) on a grid that does not fit LDS, large batches: slope records {z, m} -- one contiguous
    // run of 4 L values per query, one division per value instead of three, an item per lane (eval_slopes2d_kernel; the
    // copy, 2 x the grid, is built on first use).  NDI_SLOPES2D_KERNEL=0 / 1: A/B.
    {
      static const bool tune_live5 = std::getenv("NDI_TUNE_LIVE") != nullptr;
      static const int on_once5 = ShortKnobs::env("NDI_SLOPES2D_KERNEL", -1);
      const int on = tune_live5 ? ShortKnobs::env("NDI_SLOPES2D_KERNEL", -1) : on_once5;
      const size_t cell_b = (size_t)lanes * sizeof(T);
      // (f32 rows beyond 48 bytes: 13 .. 16 trips -- the compiler stops unrolling and the per-trip arrays go to scratch; the
      //  query-order kernel was level there anyway)
      if (on != 0 && path != NDI_PATH_BUCKETED && lanes >= 1 && cell_b <= (sizeof(T) == 4 ? 48 : 64) && nx <= 16384 && ny <= 16384 &&
          (uint64_t)slopes_bytes() < (1ull << 32) && slopes_bytes() <= SLOPES_LIMIT &&
          // AUTO: measured ahead of (f64 x 8, f32 x 16: level with) the query-order kernel on every row of up to 64 bytes
          // (profiles/r06_slopes2d_rates.txt); batches large enough to pay for building the copy; 1-2 values per point:
          // the one-thread-per-query kernel keeps the smaller batches, as before
          (on > 0 || (nq >= (lanes <= 2 ? 524288u : 65536u) && (double)nq * (double)cell_b >= 2.0 * (double)slopes_bytes()))) {
        px.ensure_dense_lut();
        py.ensure_dense_lut();
        const size_t fixed = (((size_t)(nx + LANE_SENTINELS) * sizeof(T) + 15) & ~(size_t)15) +
                             (((size_t)(ny + LANE_SENTINELS) * sizeof(T) + 15) & ~(size_t)15) +
                             px.dlut_bytes + py.dlut_bytes + (size_t)(ny - 1) * 4 * sizeof(T);
        const unsigned tb = 256u;
        size_t need = fixed + (size_t)(tb / 64) * 2 * 64 * (4 + 4 * sizeof(T));
        const size_t res_strip = (size_t)(tb / 64) * 64 * lanes * sizeof(T);      // the rows of a batch, for 16-byte stores
        const bool wide_fits = need + res_strip <= FUSED_LDS_LIMIT;
        if (wide_fits) need += res_strip;
        if (px.dense_ok && py.dense_ok && need <= FUSED_LDS_LIMIT && ensure_slopes(s)) {
          P.kind = Plan2::SLOPES2;
          P.f_tb = tb;
          P.f_lds = need;
          P.f_lds_wide = wide_fits;
          const size_t wg_per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / P.f_lds, 32 / (P.f_tb / 64)));
          P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + P.f_tb - 1) / P.f_tb, (uint64_t)cu_count() * wg_per_cu));
          g_last_path.store(NDI_PATH_GATHER);
          P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;   // fresh output: the kernel's own range test, no pre-pass
          if (P.l_check) return P;
          const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
          ProfScope ps(s, PC_LOCATE);
          hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, qx, qy, nq, px.host_knots.front(),
                             px.host_knots.back(), py.host_knots.front(), py.host_knots.back(), mode, &st->first_fail[0]);
          NDI_HIP(hipGetLastError());
          ps.done();
          return P;
        }
      }
    }
    constexpr int small2d_lanes = 2;
    constexpr int VNs = Wide<T>::N;
    const bool small2d = lanes <= 2 || (lanes <= (uint64_t)small2d_lanes && lanes % VNs != 0);
    constexpr long f2_minlanes = 1;
    constexpr long f2_minq0 = 65536;
    const bool small2d_yields = (long)lanes >= f2_minlanes && f2_minq0 >= 0 && (long)std::min<uint64_t>(nq, 1ull << 40) >= 8 * f2_minq0;
    if (small2d && !small2d_yields && both <= LDS_STAGE_LIMIT && path != NDI_PATH_BUCKETED) {
      // short trailing axes: range pre-check, then both searches + evaluation fused in one launch
      P.kind = Plan2::SMALL;
      g_last_path.store(NDI_PATH_GATHER);
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
      ProfScope ps(s, PC_LOCATE);
      hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, qx, qy, nq, px.host_knots.front(),
                         px.host_knots.back(), py.host_knots.front(), py.host_knots.back(), mode,
                         &st->first_fail[0]);
      NDI_HIP(hipGetLastError());
      ps.done();
      return P;
    }
    // Rows of fewer than 256 vectors on axes that fit LDS twice over (the reference's 100 x 100 x 5; few-channel grids),
    // batches from 65 536 queries: query order with both searches fused in (eval_fused2d_kernel) -- no (xi, yi)
    // round trip, one reciprocal per direction and query.  Grids the tile order takes (below) are left to it.
    {
      constexpr long minq = 65536;
      constexpr int VNf = Wide<T>::N;
      const bool vec = (lanes % VNf == 0) && (out_stride % VNf == 0) && aligned16(out);
      const uint64_t LVf = vec ? lanes / VNf : lanes;
      const uint64_t cell_e = pair_packed ? 2 * lanes : lanes, row_c = pair_packed ? ny - 1 : ny;
      const uint64_t grid_vecs = nx * row_c * cell_e / (vec ? VNf : 1);
      constexpr int lut_env2 = 1;
      // workgroup size and search: the combination that keeps most waves on a CU beside the staged axes (the gathers
      // are latency-bound: waves first); the bucket indices when they cost no waves, or at most half of them
      constexpr int lut2_env = -1;
      size_t lut_b = 0;
      if (lut_env2 && lut2_env != 0 && nq >= 4096 && both <= LDS_STAGE_LIMIT) {
        px.ensure_bucket_index();
        py.ensure_bucket_index();
        lut_b = px.lut_bytes + py.lut_bytes;
      }
      auto plan = [&](bool with_lut, unsigned& tb_out, size_t& lds_out) -> size_t {
        size_t best = 0;
        for (unsigned tb : {256u, 512u, 1024u}) {
          const size_t need = both + (with_lut ? lut_b : 0) + (size_t)(tb / 64) * 64 * (sizeof(uint32_t) + 6 * sizeof(T));
          if (need > LDS_STAGE_LIMIT) continue;
          const size_t waves = std::min<size_t>((160 * 1024) / need, 32 / (tb / 64)) * (tb / 64);
          if (waves > best) { best = waves; tb_out = tb; lds_out = need; }
        }
        return best;
      };
      unsigned tb0 = 256, tb1 = 256;
      size_t lds0 = 0, lds1 = 0;
      const size_t w0 = plan(false, tb0, lds0);
      const size_t w1 = lut_b ? plan(true, tb1, lds1) : 0;
      const bool lut = w1 != 0 && (lut2_env > 0 || 2 * w1 >= w0);
      const unsigned TBf = lut ? tb1 : tb0;
      const size_t lds = lut ? lds1 : lds0;
      const bool tile_candidate = path == NDI_PATH_BUCKETED || (path == NDI_PATH_AUTO && auto_tiles(nq) && lanes * sizeof(T) >= 64);
      if (minq >= 0 && (long)std::min<uint64_t>(nq, 1ull << 40) >= minq && (long)lanes >= f2_minlanes && LVf < 256 && 64 * LVf * LVf < (1ull << 32) &&
          grid_vecs < (1ull << 32) && (lut ? w1 : w0) != 0 && !tile_candidate) {
        P.kind = Plan2::FUSED2;
        P.f_vec = vec; P.f_lv = LVf; P.f_lut = lut; P.f_lds = lds; P.f_tb = TBf;
        const size_t wg_per_cu = std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 32 / (TBf / 64)));
        P.f_grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + TBf - 1) / TBf, (uint64_t)cu_count() * wg_per_cu * 4));
        g_last_path.store(NDI_PATH_GATHER);
        P.l_check = (flags & NDI_EVAL_FRESH_OUTPUT) != 0;   // fresh output: the kernel's own range test, no pre-pass
        if (P.l_check) return P;
        const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
        ProfScope ps(s, PC_LOCATE);
        hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, qx, qy, nq, px.host_knots.front(),
                           px.host_knots.back(), py.host_knots.front(), py.host_knots.back(), mode, &st->first_fail[0]);
        NDI_HIP(hipGetLastError());
        ps.done();
        return P;
      }
    }
    // BUCKETED for 2-D = tile grouping: the queries are ordered by the tile of 2^ts x 2^ts cells they fall in and
    // eval_bilinear_tiles_kernel evaluates tile by tile out of LDS, so every grid value is read from memory once
    // per tile instead of four times per query.  It pays when a tile sees several queries per cell (C3: 2.4) and
    // cannot when the batch touches less data than the tiles hold (C5's share: 0.19 queries per cell): AUTO
    // decides by queries per cell (auto_tiles).  Needs vector rows whose length divides the workgroup, the plain
    // grid layout, a tile that fits LDS and a tile histogram that fits next to the pyramids in locate2_kernel.
    constexpr int VNp = Wide<T>::N;
    const bool vec_ok_p = (lanes % VNp == 0) && (out_stride % VNp == 0) && aligned16(out);
    const uint64_t LVp = vec_ok_p ? lanes / VNp : 0;
    constexpr int lut_env = 1;   // (the bucket index: A/B settled in round 3)
    size_t both_l = both;
    const bool use_lut = lut_env && nq >= 4096 && both <= LDS_STAGE_LIMIT;
    if (use_lut) {
      px.ensure_bucket_index();
      py.ensure_bucket_index();
      if ((px.lut_bytes || py.lut_bytes) && both + px.lut_bytes + py.lut_bytes <= LDS_STAGE_LIMIT)
        both_l = both + px.lut_bytes + py.lut_bytes;
    }
    auto ntiles_of = [&](uint64_t pts, uint32_t sh) { return (uint32_t)(((pts - 1) + ((uint64_t)1 << sh) - 1) >> sh); };
    auto tile_bytes = [&](uint32_t sh) {
      const size_t s1 = ((size_t)1 << sh) + 1;
      return s1 * s1 * lanes * sizeof(T) + 5 * s1 * sizeof(T) + 16;
    };
    uint32_t ts = 0, nty = 0, nb = 0;
    bool shape_ok = false;
    if (vec_ok_p && !pair_packed && LVp >= 1 && LVp <= 1024 && 1024 % LVp == 0 && both <= LDS_STAGE_LIMIT &&
        nx < (1ull << 31) && ny < (1ull << 31) && ny * lanes < (1ull << 32) && out_stride < (1ull << 32)) {
      static const int ts_env = [] { const char* e = std::getenv("NDI_TILE_TS"); return e ? std::atoi(e) : -1; }();
      // the largest tile that leaves room for two workgroups per CU, else the largest that fits at all
      // the tile must fit the kernel's register double-buffer (6 x 1024 16-byte vectors = 96 KiB) next to 24-32 KiB of
      // static LDS; the smaller budget is tried first (shorter staging per tile)
      for (size_t budget : {(size_t)76 * 1024, (size_t)96 * 1024}) {
        for (int sh = 6; sh >= 1 && !shape_ok; --sh) {
          if (ts_env >= 0 && sh != ts_env) continue;
          const uint64_t bins = (uint64_t)ntiles_of(nx, sh) * ntiles_of(ny, sh);
          if (tile_bytes(sh) <= budget && bins <= GROUP_MAX_BINS &&
              ((both_l + 15) & ~(size_t)15) + (size_t)bins * 4 <= LDS_STAGE_LIMIT) {
            ts = (uint32_t)sh;
            nty = ntiles_of(ny, sh);
            nb = (uint32_t)bins;
            shape_ok = true;
          }
        }
        if (shape_ok) break;
      }
    }
    const bool can_tile = shape_ok && nq >= 2 && nq < 0xffffffffull;
    bool tiled = false;
    if (path == NDI_PATH_BUCKETED) tiled = can_tile;
    else if (path == NDI_PATH_AUTO) tiled = can_tile && auto_tiles(nq);
    const uint32_t sx = ts, sy = ts;
    const bool compact_records = std::is_same<T, float>::value && nx <= 65536 && ny <= 65536;
    static const int cw_env = [] { const char* e = std::getenv("NDI_TILE_CELLWORDS"); return e ? std::atoi(e) : 1; }();   // A/B
    const bool cell_words = tiled && compact_records && both <= LDS_STAGE_LIMIT && cw_env;
    g_last_path.store(tiled ? NDI_PATH_BUCKETED : NDI_PATH_GATHER);
    // Two-level grouping (tile row, then tile: coarse_scatter2d_kernel / fine_scatter2d_kernel) when the one-pass scatter
    // would leave fewer than a line's worth of records per (slice, tile) and the batch is large enough to pay for the
    // second pass.  NDI_GROUP_TWO_LEVEL=0 / 1: A/B.
    static const bool tl_live = std::getenv("NDI_TUNE_LIVE") != nullptr;
    static const int tl_once = ShortKnobs::env("NDI_GROUP_TWO_LEVEL", -1);
    const int tl_env = tl_live ? ShortKnobs::env("NDI_GROUP_TWO_LEVEL", -1) : tl_once;
    const uint32_t ntx = tiled ? nb / nty : 0;
    const bool two_level = tiled && both <= LDS_STAGE_LIMIT && ntx >= 2 && ntx <= 4096 && nty <= 4096 &&
                           (tl_env == 1 || (tl_env < 0 && nq >= (1u << 20) && (double)nq / group_blocks() / nb < 8.0));
    uint64_t slice = 0, blocks = 0;
    if (both <= LDS_STAGE_LIMIT) {   // both axes in one launch
      Locate2Args<T> LA{};
      LA.px = px.view; LA.py = py.view;
      LA.qx = qx; LA.qy = qy; LA.nq = nq;
      LA.xi = sc.idx.as<uint32_t>(); LA.yi = sc.idx2.as<uint32_t>();
      if (cell_words) LA.yi = nullptr;   // one cell word per query: all the scatter needs
      LA.first_fail = &st->first_fail[0];
      LA.mode = mode;
      LA.bx = BucketIndex<T>{nullptr, 0, T(0)};
      LA.by = LA.bx;
      if (both_l != both) {
        LA.bx = px.bidx;      // [x pyramid | y pyramid | x lut | y lut]; an axis whose formula guess is exact has none
        LA.by = py.bidx;
      }
      size_t shmem = both_l;
      if (tiled) {
        sc.hist.reserve((size_t)std::max<uint32_t>(group_blocks(), GROUP_MAX_BLOCKS) * nb * sizeof(uint32_t));
        LA.hist = sc.hist.as<uint32_t>();
        LA.nb = nb; LA.sx = sx; LA.sy = sy; LA.nty = nty;
        shmem = ((both_l + 15) & ~(size_t)15) + (size_t)nb * 4;
        if (two_level) {
          sc.chist.reserve((size_t)std::max<uint32_t>(group_blocks(), GROUP_MAX_BLOCKS) * ntx * sizeof(uint32_t));
          LA.chist = sc.chist.as<uint32_t>();
          LA.ntx = ntx;
        }
      }
      const unsigned threads = beside_eval ? 256u : threads_for_lds(shmem);
      blocks = std::max<uint64_t>(1, std::min<uint64_t>((nq + 1023) / 1024, tiled ? group_blocks() : 2048));
      if ((LA.bx.lut || LA.by.lut || tiled) && !(tiled && group_blocks() > GROUP_MAX_BLOCKS))   // staging is the fixed cost of a
        // workgroup: no more workgroups than the chip holds at once (NDI_GROUP_BLOCKS > 256 lifts the cap: A/B)
        blocks = std::min<uint64_t>(blocks, (uint64_t)cu_count() * std::max<size_t>(1, (160 * 1024) / shmem));
      slice = (nq + blocks - 1) / blocks;
      slice = (slice + threads - 1) / threads * threads;
      blocks = (nq + slice - 1) / slice;
      LA.slice = slice;
      if (two_level) {   // column-block-major tile histograms: (blocks * hist_w) words per block, ceil(nb / hist_w) <= blocks blocks
        LA.hist_w = (uint32_t)((nb + blocks - 1) / blocks);
        sc.hist.reserve((size_t)blocks * LA.hist_w * blocks * sizeof(uint32_t));
        LA.hist = sc.hist.as<uint32_t>();
      }
      allow_dynamic_lds(reinterpret_cast<const void*>(&locate2_kernel<T, LOCATE_QB>), (int)LDS_STAGE_LIMIT);
      allow_dynamic_lds(reinterpret_cast<const void*>(&locate2_kernel<T, 1>), (int)LDS_STAGE_LIMIT);
      if (slice / threads >= 16) launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate2_kernel<T, LOCATE_QB>, LA);
      else launch1<T>(s, PC_LOCATE, dim3((unsigned)blocks), dim3(threads), shmem, locate2_kernel<T, 1>, LA);
    } else {
      run_locate<T>(s, px, qx, nq, sc.idx.as<uint32_t>(), nullptr, nullptr, &st->first_fail[0], mode);
      run_locate<T>(s, py, qy, nq, sc.idx2.as<uint32_t>(), nullptr, nullptr, &st->first_fail[1], mode);
    }
    if (tiled) {
      P.kind = Plan2::TILED;
      P.ts = ts; P.nty = nty; P.nb = nb;
      P.compact = compact_records;
      // the scatter is a latency-bound chain (load -> LDS atomic -> scattered store): many waves per CU
      const unsigned gthreads = beside_eval ? 256u : (slice >= 4096 ? 1024u : (unsigned)BLOCK);
      sc.perm.reserve(nq * sizeof(uint4));            // grouped records
      if (!P.compact) sc.recq.reserve(nq * 2 * sizeof(T));
      sc.counts.reserve(((size_t)nb + 4) * sizeof(uint32_t));   // (+4: the fused scan reads / writes 16-byte pieces)
      sc.cursor.reserve(((size_t)nb + 4) * sizeof(uint32_t));
      allow_dynamic_lds(reinterpret_cast<const void*>(&group_scatter2d_kernel<T, true>), (int)(GROUP_MAX_BINS * 4));
      allow_dynamic_lds(reinterpret_cast<const void*>(&group_scatter2d_kernel<T, false>), (int)(GROUP_MAX_BINS * 4));
      ProfScope ps(s, PC_GROUP);
      const uint32_t chunk = tile_chunk();
      sc.chunkbin.reserve(((nq + chunk - 1) / chunk) * sizeof(uint32_t));
      if (!two_level)   // slice offsets, the scan of the tile totals and the tile each evaluation chunk starts in: one launch
        hipLaunchKernelGGL(group_offsets_scan_kernel, dim3((nb + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s,
                           sc.hist.as<uint32_t>(), (uint32_t)blocks, nb, sc.counts.as<uint32_t>(),
                           sc.cursor.as<uint32_t>(), st, nq, chunk, sc.chunkbin.as<uint32_t>());
      const uint32_t* yi_in = cell_words ? (const uint32_t*)nullptr : (const uint32_t*)sc.idx2.as<uint32_t>();
      if (two_level) {
        sc.perm2.reserve(nq * sizeof(uint4));
        if (!P.compact) sc.recq2.reserve(nq * 2 * sizeof(T));
        sc.cursor2.reserve(((size_t)nb + 4) * sizeof(uint32_t));
        const size_t shm_c = ((size_t)3 * ntx + (size_t)2 * gthreads) * 4;
        static const int ft_env = [] { const char* e = std::getenv("NDI_GROUP_FINE_THREADS"); return e ? std::atoi(e) : 0; }();
        constexpr int fg_env = 0;
        constexpr int FR = 4;   // records per thread and round of the fine pass (2 / 8 measured level: r05_c3_fine_round_variants.txt)
        const unsigned fthreads = (ft_env == 256 || ft_env == 512 || ft_env == 1024) ? (unsigned)ft_env : 1024u;
        // parts per tile row: one round per workgroup on evenly spread queries (measured at C3, profiles/r05_tuning.md:
        // 2560 one-round workgroups 78 us; 512 workgroups walking five rounds each with the next round's records in
        // flight 115 us -- the pass wants its parallelism across workgroups)
        const uint64_t per_row = (nq + ntx - 1) / ntx;
        const uint64_t rounds_row = (per_row + (uint64_t)FR * fthreads - 1) / ((uint64_t)FR * fthreads);
        uint32_t G = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rounds_row, 4096));
        if (fg_env > 0) G = (uint32_t)std::min<int>(fg_env, 4096);
        const unsigned fgrid = (unsigned)(((ntx + 7u) / 8u) * 8u * G);
        if (std::getenv("NDI_TRACE_PLAN"))
          std::fprintf(stderr, "[ndi plan] two-level grouping ntx=%u nty=%u slices=%llu G=%u\n", ntx, nty,
                       (unsigned long long)blocks, G);
        const uint32_t* yi_c = P.compact ? yi_in : (const uint32_t*)sc.idx2.as<uint32_t>();
        T* rq2 = P.compact ? (T*)nullptr : sc.recq2.as<T>();
        T* rq1 = P.compact ? (T*)nullptr : sc.recq.as<T>();
#define NDI_COARSE(CP)                                                                                              \
  do {                                                                                                              \
    allow_dynamic_lds(reinterpret_cast<const void*>(&coarse_scatter2d_kernel<T, CP>), (int)(64 * 1024));            \
    hipLaunchKernelGGL((coarse_scatter2d_kernel<T, CP>), dim3((unsigned)blocks), dim3(gthreads), shm_c, s,          \
                       (const uint32_t*)sc.idx.as<uint32_t>(), yi_c, qx, qy, nq, slice,                             \
                       (const uint32_t*)sc.chist.as<uint32_t>(), (const uint32_t*)sc.hist.as<uint32_t>(), nb, ntx,  \
                       sx, sc.perm2.as<uint4>(), rq2, sc.counts.as<uint32_t>());                                    \
  } while (0)
#define NDI_FINE(CP, R)                                                                                             \
  hipLaunchKernelGGL((fine_scatter2d_kernel<T, CP, R>), dim3(fgrid), dim3(fthreads), (size_t)nty * 8, s,            \
                     (const uint4*)sc.perm2.as<uint4>(), (const T*)rq2, sc.perm.as<uint4>(), rq1,                   \
                     (const uint32_t*)sc.cursor.as<uint32_t>(), sc.cursor2.as<uint32_t>(), nq, ntx, nty, sy, G)
        static const int csort_env = [] { const char* e = std::getenv("NDI_GROUP_COARSE_SORT"); return e ? std::atoi(e) : 1; }();
        const size_t shm_cs = ((((size_t)6 * ntx + 2 * gthreads) * 4 + 15) & ~(size_t)15) + (size_t)4 * gthreads * 20;
        if (P.compact && std::is_same<T, float>::value && csort_env && shm_cs <= 150 * 1024) {   // trips sorted in LDS (NDI_GROUP_COARSE_SORT=0: direct, A/B)
          auto kern = coarse_scatter2d_kernel<T, true, true>;
          allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(150 * 1024));
          hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(gthreads), shm_cs, s,
                             (const uint32_t*)sc.idx.as<uint32_t>(), yi_c, qx, qy, nq, slice,
                             (const uint32_t*)sc.chist.as<uint32_t>(), (const uint32_t*)sc.hist.as<uint32_t>(), nb, ntx,
                             sx, sc.perm2.as<uint4>(), rq2, sc.counts.as<uint32_t>());
        } else if (P.compact) NDI_COARSE(true); else NDI_COARSE(false);
        hipLaunchKernelGGL(scan_bin_totals_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)sc.counts.as<uint32_t>(), nb,
                           sc.cursor.as<uint32_t>(), sc.cursor2.as<uint32_t>(), st, chunk, sc.chunkbin.as<uint32_t>());
        static const int fsort_env = [] { const char* e = std::getenv("NDI_GROUP_FINE_SORT"); return e ? std::atoi(e) : 1; }();
        if (P.compact && std::is_same<T, float>::value && fsort_env) {   // the round's records sorted in LDS, coalesced copy-out (NDI_GROUP_FINE_SORT=0: the direct form, A/B)
#define NDI_FSORT(SR, STB)                                                                                          \
  do {                                                                                                              \
    const uint64_t rr = (per_row + (uint64_t)SR * STB - 1) / ((uint64_t)SR * STB);                                  \
    const uint32_t Gs = fg_env > 0 ? G : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rr, 4096));             \
    const unsigned sgrid = (unsigned)(((ntx + 7u) / 8u) * 8u * Gs);                                                 \
    const size_t shm_s = ((((size_t)3 * nty + 16) * 4 + 15) & ~(size_t)15) + (size_t)SR * STB * 20;                 \
    auto kern = fine_scatter2d_sorted_kernel<SR, STB>;                                                              \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(128 * 1024));                                      \
    hipLaunchKernelGGL(kern, dim3(sgrid), dim3(STB), shm_s, s, (const uint4*)sc.perm2.as<uint4>(), sc.perm.as<uint4>(), \
                       (const uint32_t*)sc.cursor.as<uint32_t>(), sc.cursor2.as<uint32_t>(), nq, ntx, nty, sy, Gs); \
  } while (0)
          if (fsort_env == 2) NDI_FSORT(8, 512); else if (fsort_env == 3) NDI_FSORT(4, 1024); else if (fsort_env == 4) NDI_FSORT(2, 512);
          else if (fsort_env == 5) NDI_FSORT(4, 256); else NDI_FSORT(4, 512);
#undef NDI_FSORT
        } else if (P.compact) NDI_FINE(true, FR);
        else NDI_FINE(false, FR);
#undef NDI_COARSE
#undef NDI_FINE
      } else if (P.compact)
        hipLaunchKernelGGL((group_scatter2d_kernel<T, true>), dim3((unsigned)blocks), dim3(gthreads), (size_t)nb * 4, s,
                           (const uint32_t*)sc.idx.as<uint32_t>(), yi_in, qx, qy, nq,
                           slice, (const uint32_t*)sc.hist.as<uint32_t>(), (const uint32_t*)sc.cursor.as<uint32_t>(),
                           nb, sx, sy, nty, sc.perm.as<uint4>(), (T*)nullptr);
      else
        hipLaunchKernelGGL((group_scatter2d_kernel<T, false>), dim3((unsigned)blocks), dim3(gthreads), (size_t)nb * 4, s,
                           (const uint32_t*)sc.idx.as<uint32_t>(), (const uint32_t*)sc.idx2.as<uint32_t>(), qx, qy, nq,
                           slice, (const uint32_t*)sc.hist.as<uint32_t>(), (const uint32_t*)sc.cursor.as<uint32_t>(),
                           nb, sx, sy, nty, sc.perm.as<uint4>(), sc.recq.as<T>());
      NDI_HIP(hipGetLastError());
      ps.done();
    }
    return P;
  }

  // grouped positions per unit of work of eval_bilinear_tiles_kernel (sweep: profiles/r03_c3_grouped.md)
  static uint32_t tile_chunk() {
    static const int chunk_env = [] { const char* e = std::getenv("NDI_TILE_CHUNK"); return e ? std::atoi(e) : 0; }();
    return chunk_env > 0 ? (uint32_t)chunk_env : 8192u;
  }

  // AUTO for 2-D: tile-grouped order when the batch has enough queries per grid cell to pay for staging every
  // tile once (C3 grid, profiles/r04_c3_qpc_sweep.jsonl: 0.95 queries per cell +5 % time, 1.2: -12 %, 1.4: -23 %,
  // 1.9: -34 %, 2.4 (C3): -37 %; below 1 the gather order wins) and the grid is far larger than what the caches hold.
  bool auto_tiles(uint64_t nq) const {
    constexpr double thr = 1.1;
    const double cells = (double)(nx - 1) * (double)(ny - 1);
    const size_t grid_bytes = (size_t)nx * ny * lanes * sizeof(T);
    if ((double)nq >= thr * cells && grid_bytes >= ((size_t)256 << 20) && lanes * sizeof(T) >= 64) return true;
    // Smaller grids (they sit in L2 / the Infinity Cache, the gather order is not HBM-bound) with rows of 256 bytes and
    // more: the tile order still wins -- up to 2 x -- once the batch is large enough to cover its fixed cost of staging
    // every tile (~0.13 ms) and has two queries per cell (profiles/r06_tiles_small_grids.jsonl: 64 MB grid, 1e6 queries
    // 0.206 -> 0.161 ms; 125 MB, 2e6: 0.38 -> 0.22; at 1 query per cell or 10 MB grids the gather order stays ahead).
    // Found by tests/test_gpu_auto_guard.py.
    return (double)nq >= 2.0 * cells && nq >= 800000 && lanes * sizeof(T) >= 256 && (lanes * sizeof(T)) % 16 == 0 && !pair_packed &&
           grid_bytes >= ((size_t)20 << 20);
  }

  void launch_eval(hipStream_t s, Scratch& sc, const Plan2& P) {
    StatusBlock* st = sc.status.as<StatusBlock>();
    const uint64_t nq = P.nq;
    if (P.kind == Plan2::SMALL) {
      const size_t both = ((px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15);
      const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
      allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small2d_kernel<T>), (int)LDS_STAGE_LIMIT);
      EvalSmall2Args<T> S{};
      S.px = px.view; S.py = py.view;
      S.data = data.as<T>();
      S.qx = P.qx; S.qy = P.qy;
      S.out = P.out;
      S.nq = nq;
      S.out_stride = P.out_stride;
      S.row_cells = pair_packed ? ny - 1 : ny;
      S.cell_elems = pair_packed ? 2 * lanes : lanes;
      S.lanes = (uint32_t)lanes;
      S.mode = mode;
      S.first_fail = &st->first_fail[0];
      S.prechecked = 1;
      // large batches: bucket indices behind the pyramids (as the two-axis search stages them), and from two values per
      // query the shared-divisor division (two reciprocals per query instead of three divisions per value: same bits)
      size_t shm = both;
      constexpr int lut_env = 1;   // (the bucket index: A/B settled in round 3)
      if (lut_env && nq >= 4096) {
        px.ensure_bucket_index();
        py.ensure_bucket_index();
        if ((px.lut_bytes || py.lut_bytes) && both + px.lut_bytes + py.lut_bytes <= LDS_STAGE_LIMIT / 2) {
          S.bx = px.bidx;
          S.by = py.bidx;
          shm = both + px.lut_bytes + py.lut_bytes;
        }
      }
      S.sdiv = lanes >= 2 ? 1 : 0;
      unsigned gs = g;
      if (shm > both)   // staging is the fixed cost of a workgroup: no more workgroups than the chip holds at once
        gs = (unsigned)std::min<uint64_t>(g, (uint64_t)cu_count() * std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / shm)));
      launch1<T>(s, PC_EVAL, dim3(gs), dim3(BLOCK), shm, eval_small2d_kernel<T>, S);
      return;
    }
    if (P.kind == Plan2::SLOPES2) {
      EvalSlopes2Args<T> F{};
      F.xk = px.view.lv0; F.yk = py.view.lv0;
      F.nx = (uint32_t)nx; F.ny = (uint32_t)ny;
      F.dx = px.dlut; F.dy = py.dlut;
      F.recs = slopes.as<T>();
#ifdef NDI_TUNING
      F.debug = ShortKnobs::env("NDI_SLOPES2D_DEBUG", 0);
#endif
      if (slopes_cells()) {
        F.col_bytes = (uint32_t)slopes_cell_stride_bytes();
        F.row_bytes = (uint32_t)((ny - 1) * slopes_cell_stride_bytes());
      } else {
        F.col_bytes = (uint32_t)(2 * lanes * sizeof(T));
        F.row_bytes = (uint32_t)(ny * 2 * lanes * sizeof(T));
      }
      F.qx = P.qx; F.qy = P.qy;
      F.out = P.out;
      F.nq = nq;
      F.out_stride = P.out_stride;
      F.lanes = (uint32_t)lanes;
      F.mode = mode;
      F.first_fail = &st->first_fail[0];
      F.check = P.l_check ? 1 : 0;
      if (std::getenv("NDI_TRACE_PLAN"))
        std::fprintf(stderr, "[ndi plan] slopes2d L=%llu recs=%s tb=%u grid=%u lds=%zu prepass=%d\n", (unsigned long long)lanes,
                     slopes_cells() ? "cell" : "point", P.f_tb, P.f_grid, P.f_lds, P.l_check ? 0 : 1);
      // rows leave as 16-byte vectors through a wave-private strip when the buffer allows it (+2-6 % over one value per lane)
      const int contig = P.out_stride != lanes ? 0 : (aligned16(P.out) && P.f_lds_wide) ? 2 : 1;
      const int mk = (int)std::max(px.dlut.lut ? px.dlut.maxk : 0u, py.dlut.lut ? py.dlut.maxk : 0u) <= 4 ? 4 : 8;
#define NDI_SL2K(LCS, MKS, CT)                                                            \
  do {                                                                                    \
    auto kern = eval_slopes2d_kernel<T, LCS, MKS, CT, 256>;                               \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)FUSED_LDS_LIMIT);         \
    launch1<T>(s, PC_EVAL, dim3(P.f_grid), dim3(256), P.f_lds, kern, F);                  \
  } while (0)
#define NDI_SL2(LCS)                                                                      \
  case LCS:                                                                               \
    if (mk == 4) { if (contig == 2) NDI_SL2K(LCS, 4, 2); else if (contig) NDI_SL2K(LCS, 4, 1); else NDI_SL2K(LCS, 4, 0); }    \
    else { if (contig == 2) NDI_SL2K(LCS, 8, 2); else if (contig) NDI_SL2K(LCS, 8, 1); else NDI_SL2K(LCS, 8, 0); }            \
    break
      switch ((int)lanes) {
        NDI_SL2(1); NDI_SL2(2); NDI_SL2(3); NDI_SL2(4); NDI_SL2(5); NDI_SL2(6); NDI_SL2(7); NDI_SL2(8);
        default:
          if constexpr (sizeof(T) == 4) {
            switch ((int)lanes) {
              NDI_SL2(9); NDI_SL2(10); NDI_SL2(11); NDI_SL2(12);
              default: break;
            }
          }
          break;
      }
#undef NDI_SL2K
#undef NDI_SL2
      return;
    }
    if (P.kind == Plan2::LANES2) {
      EvalLanes2Args<T> F{};
      F.xk = px.view.lv0; F.yk = py.view.lv0;
      F.nx = (uint32_t)nx; F.ny = (uint32_t)ny;
      F.dx = px.dlut; F.dy = py.dlut;
      F.data = data.as<T>();
      F.qx = P.qx; F.qy = P.qy;
      F.out = P.out;
      F.nq = nq;
      F.out_stride = P.out_stride;
      F.lanes = (uint32_t)lanes;
      F.mode = mode;
      F.first_fail = &st->first_fail[0];
      F.check = P.l_check ? 1 : 0;
      if (std::getenv("NDI_TRACE_PLAN"))
        std::fprintf(stderr, "[ndi plan] lanes2d L=%llu qpl=%d maxk=%u,%u tb=%u grid=%u lds=%zu prepass=%d\n", (unsigned long long)lanes,
                     P.l_qpl, px.dlut.maxk, py.dlut.maxk, P.f_tb, P.f_grid, P.f_lds, P.l_check ? 0 : 1);
      constexpr int VNl = Wide<T>::N;
#define NDI_L2(KERN)                                                                      \
  do {                                                                                    \
    auto kern = KERN;                                                                     \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)FUSED_LDS_LIMIT);         \
    launch1<T>(s, PC_EVAL, dim3(P.f_grid), dim3(P.f_tb), P.f_lds, kern, F);               \
  } while (0)
      if (lanes == 1 && P.l_qpl == VNl) {
        if (P.f_tb == 1024) NDI_L2((eval_scalar2d_kernel<T, VNl, 1024>)); else NDI_L2((eval_scalar2d_kernel<T, VNl, 256>));
      } else if (lanes == 1) {
        if (P.f_tb == 1024) NDI_L2((eval_scalar2d_kernel<T, 1, 1024>)); else NDI_L2((eval_scalar2d_kernel<T, 1, 256>));
      } else {
        if (P.f_tb == 1024) NDI_L2((eval_lanes2d_kernel<T, 1024>)); else NDI_L2((eval_lanes2d_kernel<T, 256>));
      }
#undef NDI_L2
      return;
    }
    if (P.kind == Plan2::FUSED2) {
      EvalFused2Args<T> F{};
      F.px = px.view; F.py = py.view;
      F.bx = P.f_lut ? px.bidx : BucketIndex<T>{nullptr, 0, T(0)};
      F.by = P.f_lut ? py.bidx : BucketIndex<T>{nullptr, 0, T(0)};
      F.data = data.as<T>();
      F.qx = P.qx; F.qy = P.qy;
      F.out = P.out;
      F.nq = nq;
      F.out_stride = P.out_stride;
      F.lanes = (uint32_t)lanes;
      F.lv = (uint32_t)P.f_lv;
      F.lv_magic = F.lv >= 2 ? (uint32_t)(((1ull << 32) + F.lv - 1) / F.lv) : 0u;
      const uint64_t vec = P.f_vec ? Wide<T>::N : 1;
      F.cell_vecs = (uint32_t)((pair_packed ? 2 * lanes : lanes) / vec);
      F.row_vecs = (uint32_t)((pair_packed ? ny - 1 : ny) * (pair_packed ? 2 * lanes : lanes) / vec);
      F.mode = mode;
      F.first_fail = &st->first_fail[0];
      F.check = P.l_check ? 1 : 0;
      if (std::getenv("NDI_TRACE_PLAN"))
        std::fprintf(stderr, "[ndi plan] fused2d vec=%d lv=%u lut=%d tb=%u grid=%u lds=%zu prepass=%d\n", (int)P.f_vec, F.lv, (int)P.f_lut,
                     P.f_tb, P.f_grid, P.f_lds, P.l_check ? 0 : 1);
      constexpr int VNf = Wide<T>::N;
#define NDI_F2(VEC, TB)                                                                      \
  do {                                                                                       \
    auto kern = eval_fused2d_kernel<T, VEC, 2, TB>;                                          \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)LDS_STAGE_LIMIT);            \
    launch1<T>(s, PC_EVAL, dim3(P.f_grid), dim3(TB), P.f_lds, kern, F);                      \
  } while (0)
      if (P.f_vec) {
        if (P.f_tb == 1024) NDI_F2(VNf, 1024); else if (P.f_tb == 512) NDI_F2(VNf, 512); else NDI_F2(VNf, 256);
      } else {
        if (P.f_tb == 1024) NDI_F2(1, 1024); else if (P.f_tb == 512) NDI_F2(1, 512); else NDI_F2(1, 256);
      }
#undef NDI_F2
      return;
    }
    Eval2Args<T> A{};
    A.xk = px.view.lv0;
    A.yk = py.view.lv0;
    A.data = data.as<T>();
    A.qx = P.qx;
    A.qy = P.qy;
    A.xi = sc.idx.as<uint32_t>();
    A.yi = sc.idx2.as<uint32_t>();
    A.out = P.out;
    A.nx = nx;
    A.ny = ny;
    A.lanes = lanes;
    A.out_stride = P.out_stride;
    A.nq = nq;
    A.row_cells = pair_packed ? ny - 1 : ny;
    A.cell_elems = pair_packed ? 2 * lanes : lanes;
    A.status = st;
    A.rec_i = nullptr;
    A.rec_q = nullptr;
    constexpr int VNt = Wide<T>::N;
    if (P.kind == Plan2::TILED) {
      A.rec_i = sc.perm.as<uint4>();
      A.rec_q = P.compact ? nullptr : sc.recq.as<T>();
      A.bin_start = sc.cursor.as<uint32_t>();
      A.nb = P.nb; A.ts = P.ts; A.nty = P.nty;
      A.chunk = tile_chunk();
      A.chunk_bin = sc.chunkbin.as<uint32_t>();
      const size_t s1 = ((size_t)1 << P.ts) + 1;
      // Channel split: two 512-thread workgroups per CU, each staging one half of the trailing axis of its tile (values +
      // x slopes), when two such half-tiles fit the CU's LDS beside 256 records each and a half-row still fills whole
      // 16-byte vectors.  NDI_TILE_SPLIT=0 keeps one 1024-thread workgroup per CU (A/B).
      static const int split_env = [] { const char* e = std::getenv("NDI_TILE_SPLIT"); return e ? std::atoi(e) : -1; }();
      static const int slope_env0 = [] { const char* e = std::getenv("NDI_TILE_SLOPES"); return e ? std::atoi(e) : 1; }();
      const uint64_t lv_full = lanes / Wide<T>::N;
      const size_t shm_half = (s1 * s1 + (s1 - 1) * s1) * (lanes / 2) * sizeof(T) + 5 * s1 * sizeof(T) + 16;
      const size_t static512s = 256 * 16 + (P.compact ? 2 : 2 * 256) * sizeof(T) + 320;
      const bool split2 = split_env != 0 && slope_env0 != 0 && lv_full >= 2 && lv_full % 2 == 0 &&
                          s1 * s1 * (lv_full / 2) <= 5 * 512 && 512 % (lv_full / 2) == 0 &&
                          2 * (shm_half + static512s) <= 160 * 1024;
      // (Four quarter-row workgroups per CU were measured slower -- 1.05-1.08 vs 0.77-0.85 ms at C3: 64-byte half-line stores,
      //  four decodes per record -- and are gone: profiles/r05_tuning.md 2.)
      A.ch_split = split2 ? 2u : 1u;
      {   // item -> (grid row, vector) of the tile staging: ceil(2^32 / vectors per tile row), full and last-column tiles
        const uint64_t lvv = lv_full / A.ch_split;
        const uint64_t full = (((uint64_t)1 << P.ts) + 1) * lvv;
        const uint64_t edge = (ny - ((uint64_t)(P.nty - 1) << P.ts)) * lvv;
        A.rvm_full = (uint32_t)((((uint64_t)1 << 32) + full - 1) / full);
        A.rvm_edge = (uint32_t)((((uint64_t)1 << 32) + edge - 1) / edge);
      }
      A.debug = 0;
#ifdef NDI_TUNING
      A.debug = ShortKnobs::env("NDI_FUSED_DEBUG", 0);
#endif
      const size_t shm = s1 * s1 * lanes * sizeof(T) + 5 * s1 * sizeof(T) + 16;
      const uint64_t nchunks = (nq + A.chunk - 1) / A.chunk;
      const uint64_t resident = (uint64_t)cu_count() * std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (shm + 64)));
      const unsigned gmul = 8u * A.ch_split;   // XCD-aware chunk order; the halves of a chunk are neighbours on one XCD
      const unsigned gx = (unsigned)((std::max<uint64_t>(1, std::min<uint64_t>(nchunks * A.ch_split, resident * 4 * A.ch_split)) + gmul - 1) / gmul * gmul);
      // One 1024-thread workgroup per CU.  NDI_TILE_WG=512 (A/B only): two 512-thread workgroups per CU when two tiles
      // (+ 256 records each) fit the 160 KiB and the tile fits the 10 x 512-vector register double buffer -- measured
      // slower at C3 (1.54 vs 1.17 ms: twice the tile staging per CU, half the rows per trip).
      static const int wg_env = [] { const char* e = std::getenv("NDI_TILE_WG"); return e ? std::atoi(e) : 0; }();
      const size_t tile_vecs = s1 * s1 * (lanes / VNt);
      const size_t static512 = 256 * 16 + (P.compact ? 2 : 2 * 256) * sizeof(T) + 320;
      const bool two_wg = wg_env == 512 && tile_vecs <= 10 * 512 && 2 * (shm + static512) <= 160 * 1024 &&
                          (lanes / VNt) <= 512 && 512 % (lanes / VNt) == 0;
      // x slopes of the tile staged next to its values (one division per channel and query instead of three): needs a
      // second tile-sized array in LDS, so the records are handed over 256 at a time.  NDI_TILE_SLOPES=0: A/B.
      static const int slope_env = [] { const char* e = std::getenv("NDI_TILE_SLOPES"); return e ? std::atoi(e) : 1; }();
      const size_t shm_slope = (s1 * s1 + (s1 - 1) * s1) * lanes * sizeof(T) + 5 * s1 * sizeof(T) + 16;
      const bool slopes = slope_env != 0 && !two_wg && shm_slope + static512 <= 160 * 1024;
      // ... 1024 at a time when that still fits (compact f32 records at C3: 143.9 KiB + 16 KiB)
      const size_t static1024 = 1024 * 16 + (P.compact ? 2 : 2 * 1024) * sizeof(T) + 320;
      const bool slopes_rb1024 = slopes && shm_slope + static1024 <= 160 * 1024 && slope_env != 256;
#define NDI_TILES(TTB, RB, MX, CP, SL)                                                                        \
  do {                                                                                                        \
    auto kern = eval_bilinear_tiles_kernel<T, VNt, TTB, RB, MX, CP, SL>;                                      \
    allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024 - RB * (16 + 2 * sizeof(T)) - 512)); \
    launch1<T>(s, PC_EVAL, dim3(gx), dim3(TTB), (SL) ? shm_slope : shm, kern, A);                             \
  } while (0)
      if (std::getenv("NDI_TRACE_PLAN"))
        std::fprintf(stderr, "[ndi plan] tiles ts=%u split=%u compact=%d grid=%u\n", P.ts, A.ch_split, (int)P.compact, gx);
      if (split2) {
        if constexpr (std::is_same<T, float>::value) {
          if (P.compact) {
            auto kern = eval_bilinear_tiles_kernel<T, VNt, 512, 256, 5, true, true>;
            allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(80 * 1024 - 256 * (16 + 2 * sizeof(T)) - 512));
            launch1<T>(s, PC_EVAL, dim3(gx), dim3(512), shm_half, kern, A);
            return;
          }
        }
        auto kern = eval_bilinear_tiles_kernel<T, VNt, 512, 256, 5, false, true>;
        allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)(80 * 1024 - 256 * (16 + 2 * sizeof(T)) - 512));
        launch1<T>(s, PC_EVAL, dim3(gx), dim3(512), shm_half, kern, A);
        return;
      }
      if constexpr (std::is_same<T, float>::value) {
        if (P.compact) {
          if (slopes_rb1024) NDI_TILES(1024, 1024, 6, true, true);
          else if (slopes) NDI_TILES(1024, 256, 6, true, true);
          else if (two_wg) NDI_TILES(512, 256, 10, true, false);
          else NDI_TILES(1024, 1024, 6, true, false);
          return;
        }
      }
      if (slopes_rb1024) NDI_TILES(1024, 1024, 6, false, true);
      else if (slopes) NDI_TILES(1024, 256, 6, false, true);
      else if (two_wg) NDI_TILES(512, 256, 10, false, false);
      else NDI_TILES(1024, 1024, 6, false, false);
#undef NDI_TILES
      return;
    }
    constexpr int VN = Wide<T>::N;
    const bool vec_ok = (lanes % VN == 0) && (P.out_stride % VN == 0) && aligned16(P.out);
    const uint64_t LV = vec_ok ? lanes / VN : lanes;
    // knots in LDS: both axes must fit next to each other with two 1024-thread workgroups per CU, and the batch
    // must be large enough to amortise the staging
    constexpr int klds_env = -1;
    const size_t knot_bytes = (size_t)(nx + ny) * sizeof(T);
    bool klds = vec_ok && knot_bytes <= 72 * 1024 && nq >= (1u << 20);
    if (klds_env >= 0) klds = klds_env != 0 && vec_ok && knot_bytes <= LDS_STAGE_LIMIT;
    ProfScope ps(s, PC_EVAL);
    if (klds) {
      constexpr int TB = 1024;
      const uint32_t tile_q = (uint32_t)std::max<uint64_t>(1, (uint64_t)TB * 2 / std::max<uint64_t>(LV, 1));
      const uint64_t ntiles = (nq + tile_q - 1) / tile_q;
      const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 512));
      // Short rows (the pair-packed layout, <= 64 bytes per grid point: C5) divide with the shared-divisor division
      // of kernels.hpp -- two IEEE reciprocals per item instead of twelve divisions: 1.5-6 % faster at C5's share
      // (A/B on two boxes: 0.773 -> 0.726 and 0.749 -> 0.736 ms; the measured ceiling of the access mix is
      // 0.72-0.74), neutral with index axes; long rows (C3, HBM-bound at 7.1 TB/s) get 3 % slower with it and keep
      // the IEEE divisions.  Same bits either way.
      allow_dynamic_lds(reinterpret_cast<const void*>(&eval_bilinear_kernel<T, VN, false, 2, TB, true>), (int)LDS_STAGE_LIMIT);
      allow_dynamic_lds(reinterpret_cast<const void*>(&eval_bilinear_kernel<T, VN, false, 2, TB, true, true>), (int)LDS_STAGE_LIMIT);
      constexpr int sdiv_env = -1;
      const bool sdiv = sdiv_env >= 0 ? sdiv_env != 0 : pair_packed;
      if (sdiv)
        hipLaunchKernelGGL((eval_bilinear_kernel<T, VN, false, 2, TB, true, true>), dim3(gx), dim3(TB), (knot_bytes + 15) & ~(size_t)15, s, A, tile_q);
      else
        hipLaunchKernelGGL((eval_bilinear_kernel<T, VN, false, 2, TB, true>), dim3(gx), dim3(TB), (knot_bytes + 15) & ~(size_t)15, s, A, tile_q);
    } else {
      const uint32_t tile_q = (uint32_t)std::max<uint64_t>(1, 1024 / std::max<uint64_t>(LV, 1));
      const uint64_t ntiles = (nq + tile_q - 1) / tile_q;
      const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 32768));
      if (vec_ok) hipLaunchKernelGGL((eval_bilinear_kernel<T, VN>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
      else hipLaunchKernelGGL((eval_bilinear_kernel<T, 1>), dim3(gx), dim3(BLOCK), 0, s, A, tile_q);
    }
    NDI_HIP(hipGetLastError());
    ps.done();
  }

  void enqueue(hipStream_t s, Workspace& ws, const T* qx, const T* qy, uint64_t nq, T* out,
               uint64_t out_stride, int path, int flags = 0) {
    launch_eval(s, ws.sc[0], prep(s, ws.sc[0], qx, qy, nq, out, out_stride, path, false, flags));
  }

  ndi_status collect(hipStream_t s, Workspace& ws, uint64_t index_offset, ndi_oob_info* info) {
    ws.ensure_status();
    NDI_HIP(hipMemcpyAsync(ws.host_status, ws.sc[0].status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
    NDI_HIP(hipStreamSynchronize(s));
    ws.pending = false;
    const unsigned long long fx = ws.host_status->first_fail[0], fy = ws.host_status->first_fail[1];
    if (fx == NO_FAIL && fy == NO_FAIL) return NDI_OK;
    return report(ws.last_q, ws.last_q2, ws.last_q_space, fx, fy, index_offset, info);
  }

  ndi_status report(const void* qx, const void* qy, int q_space, unsigned long long fx, unsigned long long fy,
                    uint64_t index_offset, ndi_oob_info* info) {
    // x is tested before y for the same query (bilinear.rs:71-80)
    const int axis = (fx <= fy) ? 0 : 1;
    const unsigned long long ff = axis == 0 ? fx : fy;
    const void* src = axis == 0 ? qx : qy;
    T v;
    if (q_space == NDI_MEM_DEVICE)
      NDI_HIP(hipMemcpy(&v, (const T*)src + ff, sizeof(T), hipMemcpyDeviceToHost));
    else
      v = ((const T*)src)[ff];
    // without extrapolation every failure is a range failure (NaN included: "x = NaN is not in range");
    // with it the only failure is the search meeting a NaN -- the query itself or an infinite query that the
    // periodic wrap turned into NaN (the reference panics: vector_extensions.rs:83-84)
    const ndi_status st = (mode != EX_NO) ? NDI_NAN_QUERY : NDI_OUT_OF_BOUNDS;
    if (info) {
      info->index = index_offset + ff;
      info->value = (double)v;
      info->axis = axis;
      info->status = st;
    }
    if (st == NDI_NAN_QUERY) return fail(st, "failed to convert NaN to usize (query %llu)", index_offset + ff);
    return fail(st, "%s = %.17g is not in range", axis == 0 ? "x" : "y", (double)v);
  }

  void stage_queries(hipStream_t s, Workspace& ws, const void* qx_, const void* qy_, uint64_t nq, int q_space,
                     const T** qx, const T** qy) {
    *qx = (const T*)qx_;
    *qy = (const T*)qy_;
    if (q_space != NDI_MEM_HOST) return;
    ws.qdev.reserve(nq * sizeof(T));
    ws.qdev2.reserve(nq * sizeof(T));
    NDI_HIP(hipMemcpyAsync(ws.qdev.p, qx_, nq * sizeof(T), hipMemcpyHostToDevice, s));
    NDI_HIP(hipMemcpyAsync(ws.qdev2.p, qy_, nq * sizeof(T), hipMemcpyHostToDevice, s));
    *qx = ws.qdev.as<T>();
    *qy = ws.qdev2.as<T>();
  }

  // Range pre-pass over a whole batch (see Interp1DImpl::enqueue_prepass): first_fail[0] = x, [1] = y.
  void enqueue_prepass(hipStream_t s, Workspace& ws, const T* qx, const T* qy, uint64_t nq) {
    ws.ensure_status();
    reset_status(ws.status.p, s);
    StatusBlock* st = ws.status.as<StatusBlock>();
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
    {
      ProfScope ps(s, PC_LOCATE);
      hipLaunchKernelGGL(range_check_kernel<T>, dim3(g), dim3(BLOCK), 0, s, qx, qy, nq, px.host_knots.front(),
                         px.host_knots.back(), py.host_knots.front(), py.host_knots.back(), mode,
                         &st->first_fail[0]);
      NDI_HIP(hipGetLastError());
      ps.done();
    }
    NDI_HIP(hipMemcpyAsync(ws.host_status, ws.status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
  }

  // Host output with short trailing axes: one fused launch per chunk (see Interp1DImpl::eval_small_host).
  ndi_status eval_small_host(hipStream_t s, Workspace& ws, const T* qx_dev, const T* qy_dev, const T* qx_orig,
                             const T* qy_orig, int q_space, uint64_t nq, T* out, uint64_t out_stride,
                             ndi_oob_info* info) {
    const uint64_t row_bytes = lanes * sizeof(T);
    const uint64_t chunk_q = std::max<uint64_t>(1, std::min<uint64_t>(nq, (64ull << 20) / row_bytes));
    constexpr size_t BOUNCE = 8ull << 20;
    ws.stage.reserve(chunk_q * row_bytes);
    ws.ensure_status();
    g_last_path.store(NDI_PATH_GATHER);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small2d_kernel<T>), (int)LDS_STAGE_LIMIT);
    StatusBlock* st = ws.sc[0].status.as<StatusBlock>();
    const size_t shmem = (px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15;
    for (uint64_t off = 0; off < nq; off += chunk_q) {
      const uint64_t cq = std::min<uint64_t>(chunk_q, nq - off);
      const size_t bytes = cq * row_bytes;
      NDI_HIP(hipMemsetAsync(st, 0xFF, 2 * sizeof(unsigned long long), s));
      EvalSmall2Args<T> A{};
      A.px = px.view; A.py = py.view;
      A.data = data.as<T>();
      A.qx = qx_dev + off; A.qy = qy_dev + off;
      A.out = ws.stage.as<T>();
      A.nq = cq;
      A.out_stride = lanes;
      A.row_cells = pair_packed ? ny - 1 : ny;
      A.cell_elems = pair_packed ? 2 * lanes : lanes;
      A.lanes = (uint32_t)lanes;
      A.mode = mode;
      A.first_fail = &st->first_fail[0];
      A.prechecked = 0;
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((cq + BLOCK - 1) / BLOCK, 4096));
      launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small2d_kernel<T>, A);
      const bool bounce = bytes <= BOUNCE;
      if (bounce) {
        ws.ensure_pin(BOUNCE);
        NDI_HIP(hipMemcpyAsync(ws.pin, ws.stage.p, bytes, hipMemcpyDeviceToHost, s));
      }
      NDI_HIP(hipMemcpyAsync(ws.host_status, st, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
      NDI_HIP(hipStreamSynchronize(s));
      const unsigned long long fx = ws.host_status->first_fail[0], fy = ws.host_status->first_fail[1];
      const unsigned long long ff = std::min(fx, fy);
      const uint64_t good = (ff == NO_FAIL) ? cq : (uint64_t)ff;
      T* dst = out + off * out_stride;
      if (good) {
        if (bounce) {
          if (out_stride == lanes) std::memcpy(dst, ws.pin, good * row_bytes);
          else
            for (uint64_t r = 0; r < good; ++r)
              std::memcpy(dst + r * out_stride, (const char*)ws.pin + r * row_bytes, row_bytes);
        } else {
          NDI_HIP(hipMemcpy2D(dst, out_stride * sizeof(T), ws.stage.p, row_bytes, row_bytes, good,
                              hipMemcpyDeviceToHost));
        }
      }
      if (ff != NO_FAIL) return report(qx_orig + off, qy_orig + off, q_space, fx, fy, off, info);
    }
    return NDI_OK;
  }

  // Zero-copy small batches, host arrays in and out (see Interp1DImpl::eval_small_zero_copy).
  static constexpr size_t ZERO_COPY_LIMIT = 1u << 20;
  bool zero_copy_fits(uint64_t nq, int q_space, int out_space) const {
    return q_space == NDI_MEM_HOST && out_space == NDI_MEM_HOST && lanes <= (uint64_t)SMALL_LANES &&
           ((px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15) <= LDS_STAGE_LIMIT &&
           nq * (lanes + 2) * sizeof(T) <= ZERO_COPY_LIMIT;
  }
  ndi_status eval_small_zero_copy(hipStream_t s, Workspace& ws, const T* qx_host, const T* qy_host, uint64_t nq,
                                  T* out, uint64_t out_stride, ndi_oob_info* info) {
    const size_t q_bytes = ((nq * sizeof(T)) + 255) & ~(size_t)255, row_bytes = lanes * sizeof(T);
    ws.ensure_pin(std::max<size_t>(2 * q_bytes + nq * row_bytes, 8ull << 20));
    ws.ensure_status();
    T* pqx = reinterpret_cast<T*>(ws.pin);
    T* pqy = reinterpret_cast<T*>((char*)ws.pin + q_bytes);
    T* po = reinterpret_cast<T*>((char*)ws.pin + 2 * q_bytes);
    std::memcpy(pqx, qx_host, nq * sizeof(T));
    std::memcpy(pqy, qy_host, nq * sizeof(T));
    g_last_path.store(NDI_PATH_GATHER);
    allow_dynamic_lds(reinterpret_cast<const void*>(&eval_small2d_kernel<T>), (int)LDS_STAGE_LIMIT);
    StatusBlock* st = ws.sc[0].status.as<StatusBlock>();
    NDI_HIP(hipMemsetAsync(st, 0xFF, 2 * sizeof(unsigned long long), s));
    EvalSmall2Args<T> A{};
    A.px = px.view; A.py = py.view;
    A.data = data.as<T>();
    A.qx = ws.pin_device<const T>(pqx); A.qy = ws.pin_device<const T>(pqy);
    A.out = ws.pin_device<T>(po);
    A.nq = nq;
    A.out_stride = lanes;
    A.row_cells = pair_packed ? ny - 1 : ny;
    A.cell_elems = pair_packed ? 2 * lanes : lanes;
    A.lanes = (uint32_t)lanes;
    A.mode = mode;
    A.first_fail = &st->first_fail[0];
    A.prechecked = 0;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nq + BLOCK - 1) / BLOCK, 4096));
    const size_t shmem = (px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15;
    launch1<T>(s, PC_EVAL, dim3(grid), dim3(BLOCK), shmem, eval_small2d_kernel<T>, A);
    NDI_HIP(hipMemcpyAsync(ws.host_status, st, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
    NDI_HIP(hipStreamSynchronize(s));
    const unsigned long long fx = ws.host_status->first_fail[0], fy = ws.host_status->first_fail[1];
    const unsigned long long ff = std::min(fx, fy);
    const uint64_t good = (ff == NO_FAIL) ? nq : (uint64_t)ff;
    if (good) {
      if (out_stride == lanes) std::memcpy(out, po, good * row_bytes);
      else
        for (uint64_t r = 0; r < good; ++r) std::memcpy(out + r * out_stride, (const char*)po + r * row_bytes, row_bytes);
    }
    if (ff != NO_FAIL) return report(qx_host, qy_host, NDI_MEM_HOST, fx, fy, 0, info);
    return NDI_OK;
  }

  ndi_status eval_body(hipStream_t s, Workspace& ws, const T* qx, const T* qy, const void* qx_orig,
                       const void* qy_orig, int q_space, uint64_t nq, void* out_, uint64_t out_stride,
                       const ndi_eval_opts& o, ndi_oob_info* info) {
    ws.last_q = qx_orig;
    ws.last_q2 = qy_orig;
    ws.last_q_space = q_space;
    ws.last_nq = nq;
    if (o.out_memspace == NDI_MEM_DEVICE) {
      enqueue(s, ws, qx, qy, nq, (T*)out_, out_stride, o.path, o.flags);
      ws.pending = true;
      if (o.async_launch) return NDI_OK;
      return collect(s, ws, 0, info);
    }
    const uint64_t row_bytes = lanes * sizeof(T);
    if (lanes <= (uint64_t)SMALL_LANES && ((px.lds_bytes + py.lds_bytes + 15) & ~(size_t)15) <= LDS_STAGE_LIMIT)
      return eval_small_host(s, ws, qx, qy, (const T*)qx_orig, (const T*)qy_orig, q_space, nq, (T*)out_, out_stride,
                             info);
    const uint64_t chunk_q = std::max<uint64_t>(1, std::min<uint64_t>(nq, (256ull << 20) / row_bytes));
    ws.stage.reserve(chunk_q * row_bytes);
    ws.ensure_status();
    for (uint64_t off = 0; off < nq; off += chunk_q) {
      const uint64_t cq = std::min<uint64_t>(chunk_q, nq - off);
      enqueue(s, ws, qx + off, qy + off, cq, ws.stage.as<T>(), lanes, o.path);
      NDI_HIP(hipMemcpyAsync(ws.host_status, ws.sc[0].status.p, sizeof(StatusBlock), hipMemcpyDeviceToHost, s));
      NDI_HIP(hipStreamSynchronize(s));
      const unsigned long long fx = ws.host_status->first_fail[0], fy = ws.host_status->first_fail[1];
      const unsigned long long ff = std::min(fx, fy);
      const uint64_t good = (ff == NO_FAIL) ? cq : (uint64_t)ff;
      if (good)
        NDI_HIP(hipMemcpy2D((T*)out_ + off * out_stride, out_stride * sizeof(T), ws.stage.p, row_bytes,
                            row_bytes, good, hipMemcpyDeviceToHost));
      if (ff != NO_FAIL)
        return report((const T*)qx_orig + off, (const T*)qy_orig + off, q_space, fx, fy, off, info);
    }
    return NDI_OK;
  }

  ndi_status eval(const void* qx_, const void* qy_, uint64_t nq, void* out_, uint64_t out_stride,
                  const ndi_eval_opts* opts, ndi_oob_info* info) override {
    DeviceGuard dg(device);
    Range rg("ndi_interp2d_eval");
    ndi_eval_opts o{};
    if (const ndi_status vs__ = take_opts(opts, o); vs__ != NDI_OK) return vs__;
    hipStream_t s = (hipStream_t)o.stream;  // NULL = the HIP default stream
    if (out_stride < lanes) return fail(NDI_BAD_ARG, "out_row_stride (%llu) < lanes (%llu)",
                                        (unsigned long long)out_stride, (unsigned long long)lanes);
    if (nq == 0) return NDI_OK;
    if (!qx_ || !qy_ || !out_) return fail(NDI_BAD_ARG, "null query / output pointer");
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    if (zero_copy_fits(nq, o.q_memspace, o.out_memspace))
      return eval_small_zero_copy(s, ws, (const T*)qx_, (const T*)qy_, nq, (T*)out_, out_stride, info);
    const T *qx, *qy;
    stage_queries(s, ws, qx_, qy_, nq, o.q_memspace, &qx, &qy);
    return eval_body(s, ws, qx, qy, qx_, qy_, o.q_memspace, nq, out_, out_stride, o, info);
  }

  ndi_status finish(void* stream, ndi_oob_info* info) override {
    DeviceGuard dg(device);
    hipStream_t s = (hipStream_t)stream;
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    if (!ws.pending) {
      NDI_HIP(hipStreamSynchronize(s));
      return NDI_OK;
    }
    return collect(s, ws, 0, info);
  }

  // ---- ring evaluation (see Interp1DImpl) ---------------------------------------------------
  struct RingRun {
    std::unique_lock<std::mutex> own;
    std::vector<void*> slots;
    uint64_t pitch = 0, chunk = 0, cq0 = 0;
    uint32_t n_slots = 0;
    Plan2 plan0;
    hipStream_t side = nullptr;
  };

  void ring_begin(hipStream_t s, Workspace& ws, const T* qx, const T* qy, uint64_t nq, const ndi_ring_desc* ring,
                  uint64_t stride, const ndi_eval_opts& o, RingRun& R) {
    R.n_slots = ring->n_slots;
    R.chunk = ring->chunk_queries;
    R.slots.resize(ring->n_slots);
    R.pitch = stride;
    if (ring->slots) {
      for (uint32_t i = 0; i < ring->n_slots; ++i) R.slots[i] = ring->slots[i];
    } else {
      R.own = std::unique_lock<std::mutex>(ring_own.mu);
      ring_own.ensure(ring->n_slots, ring->chunk_queries, stride * sizeof(T));
      for (uint32_t i = 0; i < ring->n_slots; ++i) R.slots[i] = (char*)ring_own.buf.p + (size_t)i * stride * sizeof(T);
      R.pitch = (uint64_t)ring->n_slots * stride;
    }
    R.side = ring_overlap() ? ws.side_stream() : s;
    for (Scratch& sc : ws.sc) sc.ensure_events();
    NDI_HIP(hipEventRecord(ws.order_event(), s));
    NDI_HIP(hipStreamWaitEvent(R.side, ws.order_event(), 0));
    R.cq0 = std::min<uint64_t>(R.chunk, nq);
    R.plan0 = prep(R.side, ws.sc[0], qx, qy, R.cq0, (T*)R.slots[0], R.pitch, o.path);
    NDI_HIP(hipEventRecord(ws.sc[0].prep_done, R.side));
  }

  void ring_produce(hipStream_t s, Workspace& ws, const T* qx, const T* qy, uint64_t limit, RingRun& R,
                    ndi_ring_consumer consume, void* user, const ndi_eval_opts& o, uint64_t q_offset,
                    uint32_t shard) {
    std::vector<hipEvent_t> busy(R.n_slots, nullptr);
    uint64_t k = 0;
    for (uint64_t off = 0; off < limit; off += R.chunk, ++k) {
      const uint64_t cq = std::min<uint64_t>(R.chunk, limit - off);
      const uint32_t slot = (uint32_t)(k % R.n_slots);
      Scratch& sc = ws.sc[k & 1];
      Plan2 P = R.plan0;
      if (k > 0) {
        if (k >= 2) NDI_HIP(hipStreamWaitEvent(R.side, sc.eval_done, 0));
        P = prep(R.side, sc, qx + off, qy + off, cq, (T*)R.slots[slot], R.pitch, o.path, R.side != s);
        NDI_HIP(hipEventRecord(sc.prep_done, R.side));
      }
      NDI_HIP(hipStreamWaitEvent(s, sc.prep_done, 0));
      if (busy[slot]) {
        NDI_HIP(hipStreamWaitEvent(s, busy[slot], 0));
        busy[slot] = nullptr;
      }
      launch_eval(s, sc, P);
      NDI_HIP(hipEventRecord(sc.eval_done, s));
      if (consume) {
        ndi_ring_chunk c{};
        c.index = k; c.q_begin = q_offset + off; c.q_count = cq; c.out = R.slots[slot]; c.row_stride = R.pitch;
        c.slot = slot; c.shard = shard; c.stream = (void*)s;
        busy[slot] = (hipEvent_t)consume(user, &c);
      }
    }
    NDI_HIP(hipStreamSynchronize(s));
    NDI_HIP(hipStreamSynchronize(R.side));
    for (hipEvent_t e : busy)
      if (e) NDI_HIP(hipEventSynchronize(e));
    ws.pending = false;
  }

  // Interp2D::interp_array through a device-output ring (see Interp1DImpl::eval_ring).
  ndi_status eval_ring(const void* qx_, const void* qy_, uint64_t nq, const ndi_ring_desc* ring,
                       ndi_ring_consumer consume, void* user, const ndi_eval_opts* opts,
                       ndi_oob_info* info) override {
    DeviceGuard dg(device);
    ndi_eval_opts o{};
    if (const ndi_status vs__ = take_opts(opts, o); vs__ != NDI_OK) return vs__;
    hipStream_t s = (hipStream_t)o.stream;
    uint64_t stride = 0;
    ndi_status rs = check_ring_desc(ring, lanes, &stride);
    if (rs != NDI_OK) return rs;
    if (nq == 0) return NDI_OK;
    if (!qx_ || !qy_) return fail(NDI_BAD_ARG, "null query pointer");
    Range rg("ndi_interp2d_eval_ring");
    SpaceLease lease(spaces, s);
    Workspace& ws = lease.ws;
    const T *qx, *qy;
    stage_queries(s, ws, qx_, qy_, nq, o.q_memspace, &qx, &qy);
    enqueue_prepass(s, ws, qx, qy, nq);
    RingRun R;
    ring_begin(s, ws, qx, qy, nq, ring, stride, o, R);
    NDI_HIP(hipStreamSynchronize(s));
    const unsigned long long fx = ws.host_status->first_fail[0], fy = ws.host_status->first_fail[1];
    const unsigned long long ff = std::min(fx, fy);
    const uint64_t limit = ff == NO_FAIL ? nq : std::min<uint64_t>(nq, ff);
    ring_produce(s, ws, qx, qy, limit, R, consume, user, o, 0, 0);
    if (ff == NO_FAIL) return NDI_OK;
    return report(qx_, qy_, o.q_memspace, fx, fy, 0, info);
  }

  // A replica on another (or the same) device (see Interp1DImpl::clone_to): the grid is copied device to device in
  // the layout it is kept in (plain or pair-packed).
  ndi_status clone_to(int dev, Interp2DBase** out) override {
    std::unique_ptr<Interp2DImpl<T>> h(new Interp2DImpl<T>());
    h->dtype = dtype; h->device = dev; h->lanes = lanes;
    h->mode = mode; h->nx = nx; h->ny = ny; h->pair_packed = pair_packed;
    {
      DeviceGuard dg(device);
      NDI_HIP(hipDeviceSynchronize());
    }
    DeviceGuard dg(dev);
    h->px.upload(px.host_knots.data(), nx);
    h->py.upload(py.host_knots.data(), ny);
    h->data.reserve(data.bytes);
    copy_across_devices(h->data.p, dev, data.p, device, data.bytes);
    *out = h.release();
    return NDI_OK;
  }

  // Median time of `reps` launches of probe_gather_kernel over nq queries (see kernels.hpp).
  ndi_status probe_ceiling(uint64_t nq, void* out, uint64_t out_stride, void* stream, int reps, double* ms) override {
    DeviceGuard dg(device);
    constexpr int VN = Wide<T>::N;
    if (!out || !ms || nq == 0 || reps < 1) return fail(NDI_BAD_ARG, "probe needs an output buffer, nq >= 1, reps >= 1");
    if (out_stride < lanes) return fail(NDI_BAD_ARG, "out_row_stride < lanes");
    if (lanes % VN || out_stride % VN || !aligned16(out))
      return fail(NDI_UNSUPPORTED, "the probe covers the vectorised layout only (lanes a multiple of %d)", VN);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t items = nq * (lanes / VN);
    const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((items + BLOCK - 1) / BLOCK, 32768));
    hipEvent_t a, b;
    NDI_HIP(hipEventCreate(&a));
    NDI_HIP(hipEventCreate(&b));
    std::vector<float> ts;
    for (int r = 0; r <= reps; ++r) {   // the first launch is a warm-up
      NDI_HIP(hipEventRecord(a, s));
      hipLaunchKernelGGL((probe_gather_kernel<T, VN>), dim3(g), dim3(BLOCK), 0, s, (const T*)data.as<T>(), nx, ny,
                         pair_packed ? ny - 1 : ny, pair_packed ? 2 * lanes : lanes, lanes, nq,
                         (uint64_t)(12345 + r), (T*)out, out_stride);
      NDI_HIP(hipGetLastError());
      NDI_HIP(hipEventRecord(b, s));
      NDI_HIP(hipEventSynchronize(b));
      float t = 0.f;
      NDI_HIP(hipEventElapsedTime(&t, a, b));
      if (r) ts.push_back(t);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    std::sort(ts.begin(), ts.end());
    *ms = ts[ts.size() / 2];
    return NDI_OK;
  }

  ndi_status trim() override {
    DeviceGuard dg(device);
    spaces.trim();
    {
      std::lock_guard<std::mutex> g(ring_own.mu);
      ring_own.clear();
    }
    {   // the slope-record copy (up to 256 MiB) goes too when it is complete and idle; the next batch that wants it rebuilds it
      std::lock_guard<std::mutex> g(slopes_mu);
      int st = slopes_state.load(std::memory_order_acquire);
      if (st == 3 && hipEventQuery(slopes_ev) == hipSuccess) st = 1;
      if (st == 1) {
        slopes.release();
        slopes_state.store(0, std::memory_order_release);
      } else {
        (void)hipGetLastError();
      }
    }
    return NDI_OK;
  }
};

template <class T>
static ndi_status create2d(const ndi_interp2d_desc& d, Interp2DBase** out) {
  DeviceGuard dg(d.device);
  Range rg("ndi_interp2d_create");
  std::unique_ptr<Interp2DImpl<T>> h(new Interp2DImpl<T>());
  h->dtype = d.dtype;
  h->device = d.device;
  h->mode = d.extrapolate ? EX_YES : EX_NO;
  h->nx = d.nx;
  h->ny = d.ny;
  h->lanes = d.lanes;
  std::vector<T> x = d.x ? fetch_axis<T>(d.x, d.x_len, d.memspace) : default_axis<T>(d.nx);
  std::vector<T> y = d.y ? fetch_axis<T>(d.y, d.y_len, d.memspace) : default_axis<T>(d.ny);
  const uint64_t x_len = d.x ? d.x_len : d.nx, y_len = d.y ? d.y_len : d.ny;
  if (d.validate) {
    ndi_status st = check_axes_2d<T>(x.data(), x_len, y.data(), y_len, d.nx, d.ny);
    if (st != NDI_OK) return st;
  } else if (x_len != d.nx || y_len != d.ny || d.nx < 2 || d.ny < 2) {
    return fail(NDI_BAD_ARG, "unvalidated create with inconsistent sizes");
  }
  if (d.lanes == 0) return fail(NDI_BAD_ARG, "lanes must be >= 1");
  if (d.nx > MAX_KNOTS || d.ny > MAX_KNOTS) return fail(NDI_UNSUPPORTED, "too many knots");
  if (!d.data) return fail(NDI_BAD_ARG, "null data pointer");
  h->px.upload(x.data(), d.nx);
  h->py.upload(y.data(), d.ny);
  const size_t bytes = (size_t)d.nx * d.ny * d.lanes * sizeof(T);
  const hipMemcpyKind kind = d.memspace == NDI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  // Short trailing axes (<= 64 B per grid point): keep the pair-packed layout instead of the plain one -- unless the
  // whole grid fits a CU's LDS (the reference's 100 x 100 bench grids): those stay plain, the layout eval_lanes2d_kernel
  // stages, and never leave L2 for the other kernels.  NDI_PAIR_PACK=1 packs every eligible grid (tests), 0 none.
  static const int pack_env = [] { const char* e = std::getenv("NDI_PAIR_PACK"); return e ? std::atoi(e) : -1; }();
  h->pair_packed = d.lanes * sizeof(T) <= 64 && d.ny >= 2 && (pack_env > 0 || (pack_env < 0 && bytes > FUSED_LDS_LIMIT));
  if (h->pair_packed) {
    const size_t packed = (size_t)d.nx * (d.ny - 1) * 2 * d.lanes * sizeof(T);
    h->data.reserve(packed);
    const void* src = d.data;
    DevBuf tmp;
    if (d.memspace != NDI_MEM_DEVICE) {
      tmp.reserve(bytes);
      NDI_HIP(hipMemcpy(tmp.p, d.data, bytes, kind));
      src = tmp.p;
    }
    // copy in 16-byte vectors when a grid point is a whole number of them (and both buffers are aligned)
    const size_t cell_bytes = (size_t)d.lanes * sizeof(T);
    const bool vec = cell_bytes % 16 == 0 && aligned16(src) && aligned16(h->data.p);
    const uint32_t units = vec ? (uint32_t)(cell_bytes / 16) : (uint32_t)d.lanes;
    int shift = -1;
    for (int b = 0; b < 31; ++b)
      if ((1u << b) == units) shift = b;
    const uint64_t row_out = (uint64_t)(d.ny - 1) * 2 * units;
    dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((row_out + BLOCK - 1) / BLOCK, 1024)),
              (unsigned)std::min<uint64_t>(d.nx, 16384));
    if (vec)
      hipLaunchKernelGGL(pack_pairs_kernel<uint4>, grid, dim3(BLOCK), 0, (hipStream_t) nullptr, (const uint4*)src,
                         (uint4*)h->data.p, (uint64_t)d.nx, (uint64_t)d.ny, units, shift);
    else
      hipLaunchKernelGGL(pack_pairs_kernel<T>, grid, dim3(BLOCK), 0, (hipStream_t) nullptr, (const T*)src,
                         h->data.template as<T>(), (uint64_t)d.nx, (uint64_t)d.ny, units, shift);
    NDI_HIP(hipGetLastError());
    NDI_HIP(hipDeviceSynchronize());
  } else {
    h->data.reserve(bytes);
    NDI_HIP(hipMemcpy(h->data.p, d.data, bytes, kind));
  }
  *out = h.release();
  return NDI_OK;
}

// ---------------------------------------------------------------------------------------------
// Locator: VectorExtensions::get_lower_index with the knot pyramid resident on the device
// ---------------------------------------------------------------------------------------------
struct LocatorBase {
  virtual ~LocatorBase() = default;
  int dtype = 0, device = 0;
  virtual ndi_status eval(const void* q, uint64_t nq, int64_t* out_idx, int memspace, void* stream) = 0;
  bool one_shot = false;   // built for a single search (ndi_get_lower_index_batch): no pinned buffer is set up
};

template <class T>
struct LocatorImpl final : LocatorBase {
  DevicePyramid<T> pyr;
  SpaceSet spaces;

  ndi_status eval(const void* q, uint64_t nq, int64_t* out_idx, int memspace, void* stream) override {
    DeviceGuard dg(device);
    if (nq == 0) return NDI_OK;
    Range rg("ndi_locator_eval");
    hipStream_t s = (hipStream_t)stream;
    const T* qdev = (const T*)q;
    int64_t* odev = out_idx;
    if (memspace == NDI_MEM_HOST && !one_shot && nq * (sizeof(T) + sizeof(int64_t)) <= ((size_t)1 << 20)) {
      // small host batch: zero-copy through the pinned buffer's host mapping (see eval_small_zero_copy)
      SpaceLease lease(spaces, s);
      Workspace& ws = lease.ws;
      const size_t q_bytes = ((nq * sizeof(T)) + 255) & ~(size_t)255;
      ws.ensure_pin(std::max<size_t>(q_bytes + nq * sizeof(int64_t), 8ull << 20));
      T* pq = reinterpret_cast<T*>(ws.pin);
      int64_t* po = reinterpret_cast<int64_t*>((char*)ws.pin + q_bytes);
      std::memcpy(pq, q, nq * sizeof(T));
      run_locate<T>(s, pyr, ws.pin_device<const T>(pq), nq, nullptr, ws.pin_device<int64_t>(po), nullptr, nullptr, EX_YES);
      NDI_HIP(hipStreamSynchronize(s));
      std::memcpy(out_idx, po, nq * sizeof(int64_t));
      return NDI_OK;
    }
    if (memspace == NDI_MEM_HOST) {
      SpaceLease lease(spaces, s);
      Workspace& ws = lease.ws;
      ws.qdev.reserve(nq * sizeof(T));
      ws.stage.reserve(nq * sizeof(int64_t));
      NDI_HIP(hipMemcpyAsync(ws.qdev.p, q, nq * sizeof(T), hipMemcpyHostToDevice, s));
      qdev = ws.qdev.as<T>();
      odev = ws.stage.as<int64_t>();
      run_locate<T>(s, pyr, qdev, nq, nullptr, odev, nullptr, nullptr, EX_YES);
      NDI_HIP(hipMemcpyAsync(out_idx, odev, nq * sizeof(int64_t), hipMemcpyDeviceToHost, s));
      NDI_HIP(hipStreamSynchronize(s));
      return NDI_OK;
    }
    run_locate<T>(s, pyr, qdev, nq, nullptr, odev, nullptr, nullptr, EX_YES);
    NDI_HIP(hipStreamSynchronize(s));
    return NDI_OK;
  }
};

template <class T>
static ndi_status create_locator(int device, const void* knots, uint64_t n, int memspace, LocatorBase** out) {
  DeviceGuard dg(device);
  if (n < 2) return fail(NDI_BAD_ARG, "get_lower_index needs at least 2 knots");
  if (n > MAX_KNOTS) return fail(NDI_UNSUPPORTED, "too many knots");
  std::unique_ptr<LocatorImpl<T>> h(new LocatorImpl<T>());
  h->dtype = DType<T>::id;
  h->device = device;
  std::vector<T> x = fetch_axis<T>(knots, n, memspace);
  h->pyr.upload(x.data(), n);
  *out = h.release();
  return NDI_OK;
}

// ---------------------------------------------------------------------------------------------
// sharded evaluation: one call, N handles (normally one per device), one host thread per shard
// ---------------------------------------------------------------------------------------------
// The reference's multi-worker shape is one interpolator driven from many threads over contiguous blocks of the
// query array (benches/bench_interp1d.rs:49-79).  Here shard i of n evaluates the block shard_range(nq, i, n) on
// handles[i]'s device (shard 0 on the calling thread, the others on the caller's persistent workers).  The
// reference's first-error result (interp1d/mod.rs:326-343) is reproduced across shards:
// every shard range-checks its block first, the minimum global failing index F is agreed on at a host barrier, and
// only the rows [0, F) are produced -- later rows are never written.  No device-to-device traffic.
static void shard_range(uint64_t nq, uint32_t i, uint32_t n, uint64_t* lo, uint64_t* hi) {
  const uint64_t base = nq / n, rem = nq % n;
  *lo = (uint64_t)i * base + std::min<uint64_t>(i, rem);
  *hi = *lo + base + (i < rem ? 1 : 0);
}
static uint64_t shard_lo(uint64_t nq, uint32_t i, uint32_t n) {
  uint64_t lo, hi;
  shard_range(nq, i, n, &lo, &hi);
  return lo;
}

struct ShardBarrier {
  std::mutex m;
  std::condition_variable cv;
  unsigned n, count = 0, gen = 0;
  explicit ShardBarrier(unsigned n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> l(m);
    const unsigned g = gen;
    if (++count == n) {
      count = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(l, [&] { return g != gen; });
    }
  }
  // k shards will never arrive (their worker could not be started): the others must not wait for them
  void drop(unsigned k) {
    std::unique_lock<std::mutex> l(m);
    n -= k;
    if (n > 0 && count == n) {
      count = 0;
      ++gen;
      cv.notify_all();
    }
  }
};

struct ShardOutcome {
  ndi_status st = NDI_OK;
  std::string msg;
};

// The worker threads of sharded calls are persistent and belong to the CALLING thread (thread_local): shard i of
// every sharded call a host thread makes runs on the same worker, so the per-(stream, thread) scratch of a handle is
// found again on the next call instead of being allocated and later evicted (hipMalloc / hipFree per call), and two
// host threads issuing sharded calls at the same time never wait for each other's workers (the shards of one call
// meet at a barrier: they must all be running).  Workers are joined when their owner thread exits.
class ShardWorkers {
  struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, stop = false;
  };
  std::vector<std::unique_ptr<Worker>> w_;
  static void loop(Worker* w) {
    std::unique_lock<std::mutex> l(w->m);
    for (;;) {
      w->cv.wait(l, [w] { return w->has_job || w->stop; });
      if (w->stop) return;
      std::function<void()> job = std::move(w->job);
      l.unlock();
      job();          // the shard bodies catch everything
      l.lock();
      w->has_job = false;
      w->cv.notify_all();
    }
  }

 public:
  ShardWorkers() = default;
  ShardWorkers(const ShardWorkers&) = delete;
  ShardWorkers& operator=(const ShardWorkers&) = delete;
  ~ShardWorkers() {
    for (auto& w : w_) {
      {
        std::lock_guard<std::mutex> g(w->m);
        w->stop = true;
      }
      w->cv.notify_all();
      if (w->th.joinable()) w->th.join();
    }
  }
  // A worker enters the pool only once its thread runs: if std::thread throws (the case run_shards drops shards for),
  // the thread-less Worker is taken out again, so a later call never start()s a job nobody will run.
  // NDI_TEST_FAIL_WORKER_START=k (tests only, read per call): the k-th and later workers of a pool fail to start.
  void ensure(size_t n) {
    while (w_.size() < n) {
      w_.emplace_back(new Worker());
      Worker* w = w_.back().get();
      try {
        if (const char* e = std::getenv("NDI_TEST_FAIL_WORKER_START"))
          if (*e && w_.size() >= (size_t)std::max(1, std::atoi(e))) throw std::runtime_error("injected: worker thread start");
        w->th = std::thread(loop, w);
      } catch (...) {
        w_.pop_back();
        throw;
      }
    }
  }
  size_t size() const { return w_.size(); }
  void start(size_t i, std::function<void()> f) {
    Worker* w = w_[i].get();
    {
      std::lock_guard<std::mutex> g(w->m);
      w->job = std::move(f);
      w->has_job = true;
    }
    w->cv.notify_all();
  }
  void wait(size_t n) {
    for (size_t i = 0; i < n; ++i) {
      Worker* w = w_[i].get();
      std::unique_lock<std::mutex> l(w->m);
      w->cv.wait(l, [w] { return !w->has_job; });
    }
  }
};
static ShardWorkers& shard_workers() {
  static thread_local ShardWorkers p;
  return p;
}

// Shard: constructed on the worker thread (the scratch lease is keyed by the thread), pre() = upload + range
// pre-pass (+ the ring's speculative first chunk), run(limit) = produce the shard's first `limit` rows.
template <class Shard, class Job>
static void run_shards(const Job& job, uint32_t n, std::vector<ShardOutcome>& out, unsigned long long F[2]) {
  ShardBarrier bar(n);
  std::atomic<unsigned long long> fx{NO_FAIL}, fy{NO_FAIL};
  std::atomic<int> broken{0};
  auto amin = [](std::atomic<unsigned long long>& a, unsigned long long v) {
    unsigned long long c = a.load();
    while (v < c && !a.compare_exchange_weak(c, v)) {
    }
  };
  auto work = [&](uint32_t i) {
    bool arrived = false;
    try {
      Shard c(job, i, n);
      unsigned long long lx = NO_FAIL, ly = NO_FAIL;
      c.pre(&lx, &ly);
      if (lx != NO_FAIL) amin(fx, c.lo + lx);
      if (ly != NO_FAIL) amin(fy, c.lo + ly);
      arrived = true;
      bar.wait();
      const unsigned long long f = std::min(fx.load(), fy.load());
      const uint64_t limit = broken.load() ? 0 : (f <= c.lo ? 0 : std::min<uint64_t>(c.cnt, f - c.lo));
      const ndi_status st = c.run(limit);
      if (st != NDI_OK) {
        out[i].st = st;
        out[i].msg = tls_error();
      }
    } catch (const HipFailure& f) {
      out[i].st = from_hip(f);
      out[i].msg = tls_error();
      broken.store(1);
    } catch (const std::bad_alloc&) {
      out[i].st = NDI_HIP_ERROR;
      out[i].msg = "host out of memory";
      broken.store(1);
    } catch (...) {
      out[i].st = NDI_HIP_ERROR;
      out[i].msg = "unexpected C++ exception";
      broken.store(1);
    }
    if (!arrived) bar.wait();
  };
  // The workers run `work` by reference to this frame: whatever happens while they are being started, this frame
  // must not unwind before every started worker is idle again, and the shards that did start must not wait at the
  // barrier for shards that never will.  The pool belongs to the calling thread and is busy for the whole call: a
  // nested sharded call from the same host thread (a ring consumer issuing one) is refused by the callers.
  ShardWorkers& pool = shard_workers();
  uint32_t started = 0;
  try {
    pool.ensure(n - 1);
    for (uint32_t i = 1; i < n; ++i) {
      pool.start(i - 1, [&work, i] { work(i); });
      ++started;
    }
  } catch (...) {
    broken.store(1);
    for (uint32_t i = started + 1; i < n; ++i) {
      out[i].st = NDI_HIP_ERROR;
      out[i].msg = "could not start the shard's host thread";
    }
    bar.drop(n - 1 - started);
  }
  work(0);   // shard 0 runs on the calling thread (never throws: the body catches everything)
  pool.wait(started);
  F[0] = fx.load();
  F[1] = fy.load();
}

// One sharded call at a time per calling host thread (its persistent workers are busy until the call returns).
struct ShardedCallScope {
  static bool& flag() {
    static thread_local bool in_call = false;
    return in_call;
  }
  bool nested;
  ShardedCallScope() : nested(flag()) { flag() = true; }
  ~ShardedCallScope() {
    if (!nested) flag() = false;
  }
};

template <class T>
struct Job1 {
  std::vector<Interp1DImpl<T>*> H;
  const void* q;
  uint64_t nq;
  const ndi_shard_io* io;
  uint64_t out_stride;
  const ndi_ring_desc* rings;
  ndi_ring_consumer consume;
  void* user;
  ndi_eval_opts o;
  const void* q_src(uint32_t i, uint32_t n) const {
    return (io && io[i].q) ? io[i].q : (const void*)((const T*)q + shard_lo(nq, i, n));
  }
};

template <class T>
struct Shard1 {
  const Job1<T>& J;
  uint32_t i;
  Interp1DImpl<T>* h;
  uint64_t lo, cnt;
  DeviceGuard dg;
  hipStream_t s;
  SpaceLease lease;
  Workspace& ws;
  const void* q_src;
  const T* q = nullptr;
  uint64_t stride = 0;
  typename Interp1DImpl<T>::RingRun R;
  Shard1(const Job1<T>& j, uint32_t i_, uint32_t n)
      : J(j), i(i_), h(j.H[i_]), lo(shard_lo(j.nq, i_, n)), cnt(shard_lo(j.nq, i_ + 1, n) - shard_lo(j.nq, i_, n)),
        dg(h->device), s((hipStream_t)(j.io ? j.io[i_].stream : nullptr)), lease(h->spaces, s), ws(lease.ws),
        q_src(j.q_src(i_, n)) {}
  void pre(unsigned long long* fx, unsigned long long*) {
    if (!cnt) return;
    q = h->stage_queries(s, ws, q_src, cnt, J.o.q_memspace);
    h->enqueue_prepass(s, ws, q, cnt);
    if (J.rings) {
      (void)check_ring_desc(&J.rings[i], h->lanes, &stride);   // validated before the threads started
      h->ring_begin(s, ws, q, cnt, &J.rings[i], stride, J.o, R);
    }
    NDI_HIP(hipStreamSynchronize(s));
    *fx = ws.host_status->first_fail[0];
  }
  ndi_status run(uint64_t limit) {
    if (!cnt) return NDI_OK;
    if (J.rings) {
      h->ring_produce(s, ws, q, limit, R, J.consume, J.user, J.o, lo, i);
      return NDI_OK;
    }
    if (!limit) return NDI_OK;
    ndi_eval_opts o = J.o;
    o.async_launch = 0;
    ndi_oob_info none{};
    return h->eval_body(s, ws, q, q_src, J.o.q_memspace, limit, J.io[i].out, J.out_stride, o, &none);
  }
};

template <class T>
struct Job2 {
  std::vector<Interp2DImpl<T>*> H;
  const void* qx;
  const void* qy;
  uint64_t nq;
  const ndi_shard_io* io;
  uint64_t out_stride;
  const ndi_ring_desc* rings;
  ndi_ring_consumer consume;
  void* user;
  ndi_eval_opts o;
  const void* qx_src(uint32_t i, uint32_t n) const {
    return (io && io[i].q) ? io[i].q : (const void*)((const T*)qx + shard_lo(nq, i, n));
  }
  const void* qy_src(uint32_t i, uint32_t n) const {
    return (io && io[i].q) ? io[i].qy : (const void*)((const T*)qy + shard_lo(nq, i, n));
  }
};

template <class T>
struct Shard2 {
  const Job2<T>& J;
  uint32_t i;
  Interp2DImpl<T>* h;
  uint64_t lo, cnt;
  DeviceGuard dg;
  hipStream_t s;
  SpaceLease lease;
  Workspace& ws;
  const void* qx_src;
  const void* qy_src;
  const T* qx = nullptr;
  const T* qy = nullptr;
  uint64_t stride = 0;
  typename Interp2DImpl<T>::RingRun R;
  Shard2(const Job2<T>& j, uint32_t i_, uint32_t n)
      : J(j), i(i_), h(j.H[i_]), lo(shard_lo(j.nq, i_, n)), cnt(shard_lo(j.nq, i_ + 1, n) - shard_lo(j.nq, i_, n)),
        dg(h->device), s((hipStream_t)(j.io ? j.io[i_].stream : nullptr)), lease(h->spaces, s), ws(lease.ws),
        qx_src(j.qx_src(i_, n)), qy_src(j.qy_src(i_, n)) {}
  void pre(unsigned long long* fx, unsigned long long* fy) {
    if (!cnt) return;
    h->stage_queries(s, ws, qx_src, qy_src, cnt, J.o.q_memspace, &qx, &qy);
    h->enqueue_prepass(s, ws, qx, qy, cnt);
    if (J.rings) {
      (void)check_ring_desc(&J.rings[i], h->lanes, &stride);
      h->ring_begin(s, ws, qx, qy, cnt, &J.rings[i], stride, J.o, R);
    }
    NDI_HIP(hipStreamSynchronize(s));
    *fx = ws.host_status->first_fail[0];
    *fy = ws.host_status->first_fail[1];
  }
  ndi_status run(uint64_t limit) {
    if (!cnt) return NDI_OK;
    if (J.rings) {
      h->ring_produce(s, ws, qx, qy, limit, R, J.consume, J.user, J.o, lo, i);
      return NDI_OK;
    }
    if (!limit) return NDI_OK;
    ndi_eval_opts o = J.o;
    o.async_launch = 0;
    ndi_oob_info none{};
    return h->eval_body(s, ws, qx, qy, qx_src, qy_src, J.o.q_memspace, limit, J.io[i].out, J.out_stride, o, &none);
  }
};

// Outcome of a sharded call on the calling thread: a device failure of any shard wins (lowest shard first);
// otherwise the reference's first-error result for the global index F, reported by the shard that owns it.
static ndi_status shards_failed(const std::vector<ShardOutcome>& out) {
  for (size_t i = 0; i < out.size(); ++i)
    if (out[i].st != NDI_OK) return fail(out[i].st, "shard %zu: %s", i, out[i].msg.c_str());
  return NDI_OK;
}
static uint32_t shard_owner(uint64_t nq, uint32_t n, uint64_t index) {
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t lo, hi;
    shard_range(nq, i, n, &lo, &hi);
    if (index >= lo && index < hi) return i;
  }
  return n - 1;
}

template <class T>
static ndi_status sharded1d(Job1<T>& J, ndi_oob_info* info) {
  ShardedCallScope scope;
  if (scope.nested)
    return fail(NDI_BAD_ARG, "nested sharded call: this host thread is inside a sharded evaluation (e.g. its ring "
                "consumer); issue the inner call from another thread");
  const uint32_t n = (uint32_t)J.H.size();
  std::vector<ShardOutcome> out(n);
  unsigned long long F[2];
  run_shards<Shard1<T>>(J, n, out, F);
  ndi_status st = shards_failed(out);
  if (st != NDI_OK || F[0] == NO_FAIL) return st;
  const uint32_t w = shard_owner(J.nq, n, F[0]);
  const uint64_t lo = shard_lo(J.nq, w, n);
  DeviceGuard dg(J.H[w]->device);
  return J.H[w]->report(J.q_src(w, n), J.o.q_memspace, F[0] - lo, lo, info);
}

template <class T>
static ndi_status sharded2d(Job2<T>& J, ndi_oob_info* info) {
  ShardedCallScope scope;
  if (scope.nested)
    return fail(NDI_BAD_ARG, "nested sharded call: this host thread is inside a sharded evaluation (e.g. its ring "
                "consumer); issue the inner call from another thread");
  const uint32_t n = (uint32_t)J.H.size();
  std::vector<ShardOutcome> out(n);
  unsigned long long F[2];
  run_shards<Shard2<T>>(J, n, out, F);
  ndi_status st = shards_failed(out);
  const unsigned long long f = std::min(F[0], F[1]);
  if (st != NDI_OK || f == NO_FAIL) return st;
  const uint32_t w = shard_owner(J.nq, n, f);
  const uint64_t lo = shard_lo(J.nq, w, n);
  DeviceGuard dg(J.H[w]->device);
  // indices relative to the owner's block; an axis whose first failure lies in a later shard stays larger
  const unsigned long long lx = F[0] == NO_FAIL ? NO_FAIL : F[0] - lo, ly = F[1] == NO_FAIL ? NO_FAIL : F[1] - lo;
  return J.H[w]->report(J.qx_src(w, n), J.qy_src(w, n), J.o.q_memspace, lx, ly, lo, info);
}

}  // namespace ndi

// =============================================================================================
// extern "C"
// =============================================================================================
namespace ndi {
// Checked build (-DNDI_BOUNDS): what the device-side index checks recorded since the last look (kernels.hpp,
// NDI_CHK).  Called where an entry point returns; reading the device word synchronises the device, which is fine
// for a debugging build.  A recorded violation turns any status into NDI_HIP_ERROR with the first violation's
// code, source line, index and limit.
static ndi_status bounds_verdict(ndi_status st, int device) {
#ifdef NDI_BOUNDS
  try {
    DeviceGuard dg(device);
    unsigned long long w[4] = {0, 0, 0, 0};
    NDI_HIP(hipMemcpyFromSymbol(w, HIP_SYMBOL(g_ndi_bounds), sizeof(w)));
    if (w[0] != 0) {
      const unsigned long long z[4] = {0, 0, 0, 0};
      NDI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ndi_bounds), z, sizeof(z)));
      return fail(NDI_HIP_ERROR, "device bounds check failed: %llu violation(s); first: code %u at kernels.hpp:%u, "
                  "index %llu, limit %llu", w[0], (unsigned)(w[1] >> 32), (unsigned)(w[1] & 0xffffffffu), w[2], w[3]);
    }
  } catch (const HipFailure& f) {
    return from_hip(f);
  }
#else
  (void)device;
#endif
  return st;
}
}  // namespace ndi

struct ndi_interp1d { ndi::Interp1DBase* impl; };
struct ndi_interp2d { ndi::Interp2DBase* impl; };
struct ndi_locator { ndi::LocatorBase* impl; };

#define NDI_TRY try {
#define NDI_CATCH                                                                  \
  }                                                                                \
  catch (const ndi::HipFailure& f) { return ndi::from_hip(f); }                    \
  catch (const std::bad_alloc&) { return ndi::fail(NDI_HIP_ERROR, "host out of memory"); } \
  catch (...) { return ndi::fail(NDI_HIP_ERROR, "unexpected C++ exception"); }

static ndi_status need_device(int device) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    return ndi::fail(NDI_HIP_ERROR, "no HIP device available (%s); this library has no CPU fallback",
                     e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= cnt)
    return ndi::fail(NDI_BAD_ARG, "device ordinal %d out of range [0, %d)", device, cnt);
  return NDI_OK;
}

NDI_API ndi_status ndi_interp1d_create(const ndi_interp1d_desc* desc, ndi_interp1d** out) {
  if (!desc || !out) return ndi::fail(NDI_BAD_ARG, "null argument");
  *out = nullptr;
  if (desc->dtype != NDI_F32 && desc->dtype != NDI_F64) return ndi::fail(NDI_BAD_ARG, "unknown dtype");
  if (desc->strategy != NDI_LINEAR && desc->strategy != NDI_CUBIC_SPLINE)
    return ndi::fail(NDI_BAD_ARG, "unknown strategy");
  // Builder checks that need no device come first, so they behave the same everywhere.
  if (desc->validate && desc->memspace == NDI_MEM_HOST && desc->x) {
    ndi_status st = ndi_validate1d(desc->dtype, desc->x, desc->x_len, desc->n, desc->strategy);
    if (st != NDI_OK) return st;
  } else if (desc->validate && !desc->x && desc->n < ndi::min_len_1d(desc->strategy)) {
    return ndi::fail(NDI_NOT_ENOUGH_DATA, "The chosen Interpolation strategy needs at least %llu data points",
                     (unsigned long long)ndi::min_len_1d(desc->strategy));
  }
  ndi_status ds = need_device(desc->device);
  if (ds != NDI_OK) return ds;
  NDI_TRY
  ndi::Interp1DBase* impl = nullptr;
  ndi_status st = desc->dtype == NDI_F32 ? ndi::create1d<float>(*desc, &impl)
                                         : ndi::create1d<double>(*desc, &impl);
  if (st != NDI_OK) return st;
  *out = new ndi_interp1d{impl};
  return NDI_OK;
  NDI_CATCH
}

NDI_API void ndi_interp1d_destroy(ndi_interp1d* h) {
  if (!h) return;
  try {
    ndi::DeviceGuard dg(h->impl->device);
    delete h->impl;
  } catch (...) {
  }
  delete h;
}

NDI_API ndi_status ndi_interp2d_create(const ndi_interp2d_desc* desc, ndi_interp2d** out) {
  if (!desc || !out) return ndi::fail(NDI_BAD_ARG, "null argument");
  *out = nullptr;
  if (desc->dtype != NDI_F32 && desc->dtype != NDI_F64) return ndi::fail(NDI_BAD_ARG, "unknown dtype");
  if (desc->validate && desc->memspace == NDI_MEM_HOST && desc->x && desc->y) {
    ndi_status st = ndi_validate2d(desc->dtype, desc->x, desc->x_len, desc->y, desc->y_len, desc->nx, desc->ny);
    if (st != NDI_OK) return st;
  }
  ndi_status ds = need_device(desc->device);
  if (ds != NDI_OK) return ds;
  NDI_TRY
  ndi::Interp2DBase* impl = nullptr;
  ndi_status st = desc->dtype == NDI_F32 ? ndi::create2d<float>(*desc, &impl)
                                         : ndi::create2d<double>(*desc, &impl);
  if (st != NDI_OK) return st;
  *out = new ndi_interp2d{impl};
  return NDI_OK;
  NDI_CATCH
}

NDI_API void ndi_interp2d_destroy(ndi_interp2d* h) {
  if (!h) return;
  try {
    ndi::DeviceGuard dg(h->impl->device);
    delete h->impl;
  } catch (...) {
  }
  delete h;
}

NDI_API ndi_status ndi_interp1d_clone(const ndi_interp1d* h, int32_t device, ndi_interp1d** out) {
  if (!h || !out) return ndi::fail(NDI_BAD_ARG, "null argument");
  *out = nullptr;
  ndi_status ds = need_device(device);
  if (ds != NDI_OK) return ds;
  NDI_TRY
  ndi::Range rg("ndi_interp1d_clone");
  ndi::Interp1DBase* impl = nullptr;
  ndi_status st = h->impl->clone_to(device, &impl);
  if (st != NDI_OK) return st;
  *out = new ndi_interp1d{impl};
  return NDI_OK;
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_clone(const ndi_interp2d* h, int32_t device, ndi_interp2d** out) {
  if (!h || !out) return ndi::fail(NDI_BAD_ARG, "null argument");
  *out = nullptr;
  ndi_status ds = need_device(device);
  if (ds != NDI_OK) return ds;
  NDI_TRY
  ndi::Range rg("ndi_interp2d_clone");
  ndi::Interp2DBase* impl = nullptr;
  ndi_status st = h->impl->clone_to(device, &impl);
  if (st != NDI_OK) return st;
  *out = new ndi_interp2d{impl};
  return NDI_OK;
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_coefficients(const ndi_interp1d* h, void* a_out, void* b_out, int32_t memspace) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return h->impl->coefficients(a_out, b_out, memspace);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_eval(const ndi_interp1d* h, const void* q, uint64_t nq, void* out,
                                     uint64_t out_row_stride, const ndi_eval_opts* opts, ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->eval(q, nq, out, out_row_stride, opts, info), h->impl->device);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_eval(const ndi_interp2d* h, const void* qx, const void* qy, uint64_t nq,
                                     void* out, uint64_t out_row_stride, const ndi_eval_opts* opts,
                                     ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->eval(qx, qy, nq, out, out_row_stride, opts, info), h->impl->device);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_finish(const ndi_interp1d* h, void* stream, ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->finish(stream, info), h->impl->device);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_finish(const ndi_interp2d* h, void* stream, ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->finish(stream, info), h->impl->device);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_eval_ring(const ndi_interp1d* h, const void* q, uint64_t nq,
                                          const ndi_ring_desc* ring, ndi_ring_consumer consume, void* user,
                                          const ndi_eval_opts* opts, ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->eval_ring(q, nq, ring, consume, user, opts, info), h->impl->device);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_eval_ring(const ndi_interp2d* h, const void* qx, const void* qy, uint64_t nq,
                                          const ndi_ring_desc* ring, ndi_ring_consumer consume, void* user,
                                          const ndi_eval_opts* opts, ndi_oob_info* info) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return ndi::bounds_verdict(h->impl->eval_ring(qx, qy, nq, ring, consume, user, opts, info), h->impl->device);
  NDI_CATCH
}

// ---- sharded evaluation ------------------------------------------------------------------------
NDI_API void ndi_shard_bounds(uint64_t nq, uint32_t shard, uint32_t n_shards, uint64_t* lo, uint64_t* hi) {
  uint64_t l = 0, h = 0;
  if (n_shards && shard < n_shards) ndi::shard_range(nq, shard, n_shards, &l, &h);
  if (lo) *lo = l;
  if (hi) *hi = h;
}

template <class Impl, class Handle>
static ndi_status gather_handles(const Handle* const* handles, uint32_t n, int dtype, std::vector<Impl*>& out) {
  out.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    if (handles[i]->impl->dtype != dtype || handles[i]->impl->lanes != handles[0]->impl->lanes)
      return ndi::fail(NDI_BAD_ARG, "shard %u: the handles of a sharded call must be replicas (same element type "
                       "and trailing lanes)", i);
    // ... and the same knots, strategy and extrapolation mode: every shard range-checks and evaluates with its own
    // handle, so a mismatched set would mix interpolators and report a first-error index no serial loop produces
    if (i > 0 && handles[i]->impl->signature() != handles[0]->impl->signature())
      return ndi::fail(NDI_BAD_ARG, "shard %u: the handles of a sharded call must be replicas of one interpolator "
                       "(same knots, strategy and extrapolation mode as shard 0)", i);
    out[i] = static_cast<Impl*>(handles[i]->impl);
  }
  return NDI_OK;
}

template <class Handle>
static ndi_status check_sharded_args(const Handle* const* handles, uint32_t n, const void* q, const void* qy,
                                     bool two_d, uint64_t nq, const ndi_shard_io* io, bool need_out,
                                     const ndi_ring_desc* rings, bool need_rings, uint64_t out_stride) {
  if (!handles || n == 0) return ndi::fail(NDI_BAD_ARG, "a sharded call needs at least one handle");
  for (uint32_t i = 0; i < n; ++i)
    if (!handles[i]) return ndi::fail(NDI_BAD_ARG, "shard %u: null handle", i);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < i; ++j)
      if (handles[i] == handles[j]) return ndi::fail(NDI_BAD_ARG, "shards %u and %u share one handle", j, i);
  const uint64_t lanes = handles[0]->impl->lanes;
  if (need_out && !io) return ndi::fail(NDI_BAD_ARG, "null shard io array");
  if (need_out && out_stride < lanes)
    return ndi::fail(NDI_BAD_ARG, "out_row_stride (%llu) < lanes (%llu)", (unsigned long long)out_stride,
                     (unsigned long long)lanes);
  if (need_rings && !rings) return ndi::fail(NDI_BAD_ARG, "null ring array");
  for (uint32_t i = 0; i < n && nq; ++i) {
    uint64_t lo, hi;
    ndi::shard_range(nq, i, n, &lo, &hi);
    const bool own_q = io && io[i].q;
    if (two_d && io && ((io[i].q != nullptr) != (io[i].qy != nullptr)))
      return ndi::fail(NDI_BAD_ARG, "shard %u: q and qy must be given together", i);
    if (!own_q && (!q || (two_d && !qy))) return ndi::fail(NDI_BAD_ARG, "null query pointer");
    if (need_out && hi > lo && !io[i].out) return ndi::fail(NDI_BAD_ARG, "shard %u: null output pointer", i);
    if (need_rings) {
      uint64_t stride = 0;
      ndi_status st = ndi::check_ring_desc(&rings[i], lanes, &stride);
      if (st != NDI_OK) return st;
    }
  }
  return NDI_OK;
}

NDI_API ndi_status ndi_interp1d_eval_sharded(const ndi_interp1d* const* handles, uint32_t n_shards, const void* q,
                                             uint64_t nq, const ndi_shard_io* io, uint64_t out_row_stride,
                                             const ndi_eval_opts* opts, ndi_oob_info* info) {
  ndi_status st = check_sharded_args(handles, n_shards, q, nullptr, false, nq, io, true, nullptr, false, out_row_stride);
  if (st != NDI_OK || nq == 0) return st;
  NDI_TRY
  ndi::Range rg("ndi_interp1d_eval_sharded");
  const int dtype = handles[0]->impl->dtype;
  ndi_eval_opts o{};
  if (const ndi_status vs__ = ndi::take_opts(opts, o); vs__ != NDI_OK) return vs__;
  if (dtype == NDI_F32) {
    ndi::Job1<float> J{{}, q, nq, io, out_row_stride, nullptr, nullptr, nullptr, o};
    st = gather_handles(handles, n_shards, dtype, J.H);
    return st != NDI_OK ? st : ndi::sharded1d<float>(J, info);
  }
  ndi::Job1<double> J{{}, q, nq, io, out_row_stride, nullptr, nullptr, nullptr, o};
  st = gather_handles(handles, n_shards, dtype, J.H);
  return st != NDI_OK ? st : ndi::sharded1d<double>(J, info);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_eval_ring_sharded(const ndi_interp1d* const* handles, uint32_t n_shards,
                                                  const void* q, uint64_t nq, const ndi_shard_io* io,
                                                  const ndi_ring_desc* rings, ndi_ring_consumer consume, void* user,
                                                  const ndi_eval_opts* opts, ndi_oob_info* info) {
  ndi_status st = check_sharded_args(handles, n_shards, q, nullptr, false, nq, io, false, rings, true, 0);
  if (st != NDI_OK || nq == 0) return st;
  NDI_TRY
  ndi::Range rg("ndi_interp1d_eval_ring_sharded");
  const int dtype = handles[0]->impl->dtype;
  ndi_eval_opts o{};
  if (const ndi_status vs__ = ndi::take_opts(opts, o); vs__ != NDI_OK) return vs__;
  if (dtype == NDI_F32) {
    ndi::Job1<float> J{{}, q, nq, io, 0, rings, consume, user, o};
    st = gather_handles(handles, n_shards, dtype, J.H);
    return st != NDI_OK ? st : ndi::sharded1d<float>(J, info);
  }
  ndi::Job1<double> J{{}, q, nq, io, 0, rings, consume, user, o};
  st = gather_handles(handles, n_shards, dtype, J.H);
  return st != NDI_OK ? st : ndi::sharded1d<double>(J, info);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_eval_sharded(const ndi_interp2d* const* handles, uint32_t n_shards, const void* qx,
                                             const void* qy, uint64_t nq, const ndi_shard_io* io,
                                             uint64_t out_row_stride, const ndi_eval_opts* opts,
                                             ndi_oob_info* info) {
  ndi_status st = check_sharded_args(handles, n_shards, qx, qy, true, nq, io, true, nullptr, false, out_row_stride);
  if (st != NDI_OK || nq == 0) return st;
  NDI_TRY
  ndi::Range rg("ndi_interp2d_eval_sharded");
  const int dtype = handles[0]->impl->dtype;
  ndi_eval_opts o{};
  if (const ndi_status vs__ = ndi::take_opts(opts, o); vs__ != NDI_OK) return vs__;
  if (dtype == NDI_F32) {
    ndi::Job2<float> J{{}, qx, qy, nq, io, out_row_stride, nullptr, nullptr, nullptr, o};
    st = gather_handles(handles, n_shards, dtype, J.H);
    return st != NDI_OK ? st : ndi::sharded2d<float>(J, info);
  }
  ndi::Job2<double> J{{}, qx, qy, nq, io, out_row_stride, nullptr, nullptr, nullptr, o};
  st = gather_handles(handles, n_shards, dtype, J.H);
  return st != NDI_OK ? st : ndi::sharded2d<double>(J, info);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_eval_ring_sharded(const ndi_interp2d* const* handles, uint32_t n_shards,
                                                  const void* qx, const void* qy, uint64_t nq,
                                                  const ndi_shard_io* io, const ndi_ring_desc* rings,
                                                  ndi_ring_consumer consume, void* user, const ndi_eval_opts* opts,
                                                  ndi_oob_info* info) {
  ndi_status st = check_sharded_args(handles, n_shards, qx, qy, true, nq, io, false, rings, true, 0);
  if (st != NDI_OK || nq == 0) return st;
  NDI_TRY
  ndi::Range rg("ndi_interp2d_eval_ring_sharded");
  const int dtype = handles[0]->impl->dtype;
  ndi_eval_opts o{};
  if (const ndi_status vs__ = ndi::take_opts(opts, o); vs__ != NDI_OK) return vs__;
  if (dtype == NDI_F32) {
    ndi::Job2<float> J{{}, qx, qy, nq, io, 0, rings, consume, user, o};
    st = gather_handles(handles, n_shards, dtype, J.H);
    return st != NDI_OK ? st : ndi::sharded2d<float>(J, info);
  }
  ndi::Job2<double> J{{}, qx, qy, nq, io, 0, rings, consume, user, o};
  st = gather_handles(handles, n_shards, dtype, J.H);
  return st != NDI_OK ? st : ndi::sharded2d<double>(J, info);
  NDI_CATCH
}

NDI_API ndi_status ndi_interp1d_trim(const ndi_interp1d* h) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return h->impl->trim();
  NDI_CATCH
}

NDI_API uint64_t ndi_interp1d_scratch_sets(const ndi_interp1d* h) { return h ? h->impl->scratch_sets() : 0; }

NDI_API ndi_status ndi_interp2d_trim(const ndi_interp2d* h) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return h->impl->trim();
  NDI_CATCH
}

NDI_API ndi_status ndi_interp2d_probe_ceiling(const ndi_interp2d* h, uint64_t nq, void* out, uint64_t out_row_stride,
                                              void* stream, int32_t reps, double* ms) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  NDI_TRY
  return h->impl->probe_ceiling(nq, out, out_row_stride, stream, reps, ms);
  NDI_CATCH
}

NDI_API ndi_status ndi_locator_create(int32_t dtype, int32_t device, const void* knots, uint64_t n,
                                      int32_t memspace, ndi_locator** out) {
  if (!knots || !out) return ndi::fail(NDI_BAD_ARG, "null argument");
  *out = nullptr;
  if (dtype != NDI_F32 && dtype != NDI_F64) return ndi::fail(NDI_BAD_ARG, "unknown dtype");
  ndi_status ds = need_device(device);
  if (ds != NDI_OK) return ds;
  NDI_TRY
  ndi::LocatorBase* impl = nullptr;
  ndi_status st = dtype == NDI_F32 ? ndi::create_locator<float>(device, knots, n, memspace, &impl)
                                   : ndi::create_locator<double>(device, knots, n, memspace, &impl);
  if (st != NDI_OK) return st;
  *out = new ndi_locator{impl};
  return NDI_OK;
  NDI_CATCH
}

NDI_API ndi_status ndi_locator_eval(const ndi_locator* h, const void* q, uint64_t nq, int64_t* out_idx,
                                    int32_t memspace, void* stream) {
  if (!h) return ndi::fail(NDI_BAD_ARG, "null handle");
  if ((!q || !out_idx) && nq) return ndi::fail(NDI_BAD_ARG, "null argument");
  NDI_TRY
  return h->impl->eval(q, nq, out_idx, memspace, stream);
  NDI_CATCH
}

NDI_API void ndi_locator_destroy(ndi_locator* h) {
  if (!h) return;
  try {
    ndi::DeviceGuard dg(h->impl->device);
    delete h->impl;
  } catch (...) {
  }
  delete h;
}

// One-shot form: builds a locator, searches, drops it (allocation + knot upload per call).
NDI_API ndi_status ndi_get_lower_index_batch(int32_t dtype, int32_t device, const void* knots, uint64_t n,
                                             const void* q, uint64_t nq, int64_t* out_idx, int32_t memspace) {
  if (!knots || (!q && nq) || (!out_idx && nq)) return ndi::fail(NDI_BAD_ARG, "null argument");
  ndi_locator* loc = nullptr;
  ndi_status st = ndi_locator_create(dtype, device, knots, n, memspace, &loc);
  if (st != NDI_OK) return st;
  loc->impl->one_shot = true;   // pinning a buffer costs more than the two small copies it would save
  st = ndi_locator_eval(loc, q, nq, out_idx, memspace, nullptr);
  ndi_locator_destroy(loc);
  return st;
}

NDI_API int32_t ndi_monotonic_prop(int32_t dtype, const void* host_v, uint64_t n) {
  if (dtype == NDI_F32) return ndi::monotonic_scan<float>((const float*)host_v, n);
  return ndi::monotonic_scan<double>((const double*)host_v, n);
}

NDI_API ndi_status ndi_validate1d(int32_t dtype, const void* host_x, uint64_t x_len, uint64_t n, int32_t strategy) {
  if (dtype == NDI_F32) return ndi::check_axis_1d<float>((const float*)host_x, x_len, n, strategy);
  if (dtype == NDI_F64) return ndi::check_axis_1d<double>((const double*)host_x, x_len, n, strategy);
  return ndi::fail(NDI_BAD_ARG, "unknown dtype");
}

NDI_API ndi_status ndi_validate2d(int32_t dtype, const void* host_x, uint64_t x_len, const void* host_y,
                                  uint64_t y_len, uint64_t nx, uint64_t ny) {
  if (dtype == NDI_F32)
    return ndi::check_axes_2d<float>((const float*)host_x, x_len, (const float*)host_y, y_len, nx, ny);
  if (dtype == NDI_F64)
    return ndi::check_axes_2d<double>((const double*)host_x, x_len, (const double*)host_y, y_len, nx, ny);
  return ndi::fail(NDI_BAD_ARG, "unknown dtype");
}

// ---- library-owned output buffers -------------------------------------------------------------------------------
// The rate at which a kernel streams rows into a multi-gigabyte buffer depends on which physical pages back it: the same
// evaluation runs 4.6 .. 6.1 ms per 1e6 queries (C2) into buffers the allocator hands out one after the other, stable per
// buffer whatever the order or the warm-up (profiles/r06_placement_vs_ramp.jsonl), and neither hipMemCreate chunks nor
// chunks spread over the physical range change the odds (profiles/r06_output_alloc_probe*.jsonl).  What user space CAN do
// is look: the zero fill Array::zeros performs anyway is timed, and a buffer that fills slowly is set aside and another
// one is asked for while it is still held (so the allocator cannot hand the same pages back), up to `max_tries`; the
// fastest-filling candidate is kept, the others are freed.  A sequential fill and the scattered row stream of the
// evaluation run at the same rate into a given buffer (profiles/r03_tuning.md, "Output placement").
namespace ndi {
struct OutputRegistry {
  std::mutex mu;
  std::unordered_map<void*, std::pair<int, size_t>> live;   // ptr -> (device, bytes)
  // Freed buffers kept for the next request of the same size on the same device (a caller that evaluates batch after
  // batch through interp_array would otherwise pay the allocator -- seconds for tens of gigabytes -- on every call, where a
  // caching allocator pays it once): at most two, and together at most a third of the device's memory; ndi_output_trim
  // releases them.  Their fill rate is known, so a cached buffer is taken without another search.
  struct Spare { void* p; int dev; size_t bytes; double rate; };
  std::vector<Spare> spare;
  std::unordered_map<void*, double> rate;                    // fill rate of the live buffers (TB/s)
  void rates_set(void* p, double r) {
    std::lock_guard<std::mutex> g(mu);
    rate[p] = r;
  }
};
static OutputRegistry& output_registry() {
  static OutputRegistry r;
  return r;
}
static double timed_zero_fill(void* p, size_t bytes, hipEvent_t a, hipEvent_t b) {
  const uint64_t nvec = bytes / 16;
  const uint64_t nrows = (nvec + 2047) / 2048;
  uint64_t mult = 2654435761ull % (nrows ? nrows : 1);         // a multiplier coprime to nrows: the walk is a permutation of the rows
  if (mult < 2) mult = 1;
  auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
  while (mult > 1 && gcd(mult, nrows) != 1) --mult;
  const unsigned gr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nrows, (uint64_t)cu_count() * 32));
  NDI_HIP(hipEventRecord(a, nullptr));
  hipLaunchKernelGGL(zero_fill_kernel, dim3(gr), dim3(BLOCK), 0, (hipStream_t) nullptr, (dbl2*)p, nvec, nrows, mult);
  NDI_HIP(hipGetLastError());
  if (bytes % 16) NDI_HIP(hipMemsetAsync((char*)p + nvec * 16, 0, bytes % 16, nullptr));
  NDI_HIP(hipEventRecord(b, nullptr));
  NDI_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  NDI_HIP(hipEventElapsedTime(&ms, a, b));
  return (double)ms;
}
}  // namespace ndi

NDI_API ndi_status ndi_output_alloc(int32_t device, uint64_t bytes, uint32_t max_tries, uint32_t flags, void** out,
                                    ndi_output_info* info) {
  if (!out) return ndi::fail(NDI_BAD_ARG, "null argument");
  if (flags & ~(uint32_t)NDI_OUTPUT_UNINITIALIZED) return ndi::fail(NDI_BAD_ARG, "unknown ndi_output_flags bit");
  *out = nullptr;
  if (info) *info = ndi_output_info{};
  if (bytes == 0) return ndi::fail(NDI_BAD_ARG, "zero-sized output");
  NDI_TRY
  ndi::DeviceGuard dg(device);
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  static const double accept_tbs = [] { const char* e = std::getenv("NDI_OUTPUT_ACCEPT_TBPS"); return e ? std::atof(e) : 6.6; }();
  static const int tries_env = ndi::ShortKnobs::env("NDI_OUTPUT_TRIES", 0);
  uint32_t tries = max_tries;
  if (tries == 0) {       // as many candidates as fit into about half of what is free, 2 .. 8
    size_t fr = 0, tot = 0;
    NDI_HIP(hipMemGetInfo(&fr, &tot));
    tries = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(2, (uint64_t)(0.55 * (double)fr / (double)bytes)));
  }
  if (tries_env > 0) tries = (uint32_t)tries_env;
  // small buffers: a fill this short says nothing about placement, and placement matters little to them
  if (bytes < ((uint64_t)1 << 30)) tries = 1;
  hipEvent_t ea = nullptr, eb = nullptr;
  NDI_HIP(hipEventCreate(&ea));
  NDI_HIP(hipEventCreate(&eb));
  struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ea, eb};
  {   // a kept buffer of this size on this device: zero it again and hand it out
    auto& R = ndi::output_registry();
    void* hit = nullptr;
    double rate = 0.0;
    {
      std::lock_guard<std::mutex> g(R.mu);
      for (size_t i = 0; i < R.spare.size(); ++i)
        if (R.spare[i].dev == device && R.spare[i].bytes == (size_t)bytes) {
          hit = R.spare[i].p;
          rate = R.spare[i].rate;
          R.spare.erase(R.spare.begin() + (long)i);
          break;
        }
    }
    if (hit) {
      // NDI_OUTPUT_UNINITIALIZED: the caller overwrites every row it reads (interp_array: the buffer is dropped on Err), so a
      // kept buffer -- whose placement is known -- goes out as it is; the zero fill of 32.8 GB costs what evaluating into it costs
      const double ms = (flags & NDI_OUTPUT_UNINITIALIZED) ? 0.0 : ndi::timed_zero_fill(hit, bytes, ea, eb);
      {
        std::lock_guard<std::mutex> g(R.mu);
        R.live[hit] = {device, (size_t)bytes};
      }
      R.rates_set(hit, rate);
      if (info) {
        info->tries = 0;                                         // 0: a kept buffer, no candidate was allocated
        info->fill_tbps = ms > 0.0 ? (double)bytes / (ms * 1e-3) / 1e12 : rate;   // (not refilled: the rate it was kept with)
        info->worst_fill_tbps = info->fill_tbps;
        info->alloc_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
      }
      *out = hit;
      return NDI_OK;
    }
  }
  std::vector<std::pair<void*, double>> cand;   // (pointer, TB/s of its zero fill)
  auto free_all_but = [&](void* keep) {
    for (auto& c : cand)
      if (c.first != keep) (void)hipFree(c.first);
  };
  void* best = nullptr;
  double best_rate = 0.0;
  try {
    for (uint32_t t = 0; t < tries; ++t) {
      if (t > 0) {     // a further candidate only while the device has room for it beside what is held
        size_t fr = 0, tot = 0;
        NDI_HIP(hipMemGetInfo(&fr, &tot));
        if (fr < bytes + ((size_t)2 << 30)) break;
      }
      void* p = nullptr;
      const hipError_t me = hipMalloc(&p, bytes);
      if (me != hipSuccess) {
        (void)hipGetLastError();
        if (cand.empty()) throw ndi::HipFailure{me, "hipMalloc(output buffer)", __LINE__};
        break;
      }
      double ms = ndi::timed_zero_fill(p, bytes, ea, eb);
      if (tries > 1 && t == 0) ms = ndi::timed_zero_fill(p, bytes, ea, eb);   // (the first fill of a process also pays clock ramp-up)
      const double rate = (double)bytes / (ms * 1e-3) / 1e12;
      cand.emplace_back(p, rate);
      if (rate > best_rate) { best_rate = rate; best = p; }
      if (rate >= accept_tbs) break;
    }
  } catch (...) {
    free_all_but(nullptr);
    throw;
  }
  free_all_but(best);
  {
    auto& R = ndi::output_registry();
    std::lock_guard<std::mutex> g(R.mu);
    R.live[best] = {device, (size_t)bytes};
    R.rate[best] = best_rate;
  }
  if (info) {
    info->tries = (uint32_t)cand.size();
    info->fill_tbps = best_rate;
    info->alloc_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    info->worst_fill_tbps = best_rate;
    for (auto& c : cand) info->worst_fill_tbps = std::min(info->worst_fill_tbps, c.second);
  }
  *out = best;
  return NDI_OK;
  NDI_CATCH
}

NDI_API ndi_status ndi_output_free(void* p) {
  if (!p) return NDI_OK;
  NDI_TRY
  int dev = -1;
  size_t bytes = 0;
  double rate = 0.0;
  void* evict = nullptr;
  int evict_dev = -1;
  {
    auto& R = ndi::output_registry();
    std::lock_guard<std::mutex> g(R.mu);
    auto it = R.live.find(p);
    if (it == R.live.end()) return ndi::fail(NDI_BAD_ARG, "pointer was not returned by ndi_output_alloc");
    dev = it->second.first;
    bytes = it->second.second;
    R.live.erase(it);
    auto ir = R.rate.find(p);
    if (ir != R.rate.end()) { rate = ir->second; R.rate.erase(ir); }
  }
  ndi::DeviceGuard dg(dev);
  // keep it for the next request of this size?  (buffers of >= 1 GiB only; two at most, a third of the device's memory)
  static const int keep_env = ndi::ShortKnobs::env("NDI_OUTPUT_KEEP", 2);
  bool kept = false;
  if (keep_env > 0 && bytes >= ((size_t)1 << 30)) {
    size_t fr = 0, tot = 0;
    NDI_HIP(hipMemGetInfo(&fr, &tot));
    auto& R = ndi::output_registry();
    std::lock_guard<std::mutex> g(R.mu);
    size_t held = bytes;
    for (auto& sp : R.spare)
      if (sp.dev == dev) held += sp.bytes;
    if (held <= tot / 3) {
      if (R.spare.size() >= (size_t)keep_env) {     // the oldest goes
        evict = R.spare.front().p;
        evict_dev = R.spare.front().dev;
        R.spare.erase(R.spare.begin());
      }
      R.spare.push_back({p, dev, bytes, rate});
      kept = true;
    }
  }
  // hipFree waits for the device; a kept buffer gives the same guarantee -- whoever gets it next (possibly without a refill,
  // NDI_OUTPUT_UNINITIALIZED) must not meet the previous owner's kernels still writing into it from a non-blocking stream
  if (kept) NDI_HIP(hipDeviceSynchronize());
  if (evict) {
    ndi::DeviceGuard de(evict_dev);
    NDI_HIP(hipFree(evict));
  }
  if (!kept) NDI_HIP(hipFree(p));
  return NDI_OK;
  NDI_CATCH
}

NDI_API ndi_status ndi_output_trim(void) {
  NDI_TRY
  std::vector<ndi::OutputRegistry::Spare> gone;
  {
    auto& R = ndi::output_registry();
    std::lock_guard<std::mutex> g(R.mu);
    gone.swap(R.spare);
  }
  for (auto& sp : gone) {
    ndi::DeviceGuard dg(sp.dev);
    NDI_HIP(hipFree(sp.p));
  }
  return NDI_OK;
  NDI_CATCH
}

NDI_API int32_t ndi_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

NDI_API const char* ndi_last_error_string(void) { return ndi::tls_error().c_str(); }

NDI_API uint32_t ndi_version(void) { return (NDI_VERSION_MAJOR << 16) | NDI_VERSION_MINOR; }

NDI_API void ndi_profile_enable(int32_t on) { ndi::g_prof_on.store(on ? 1 : 0); }

NDI_API ndi_status ndi_profile_read(ndi_profile* out, int32_t reset) {
  if (!out) return ndi::fail(NDI_BAD_ARG, "null argument");
  NDI_TRY
  std::lock_guard<std::mutex> g(ndi::g_prof_mu);
  for (auto& r : ndi::g_prof_recs) {
    NDI_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    NDI_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    ndi::ProfScope::account_locked(r, ms);
    ndi::g_prof_pool[r.dev].push_back(r.a);
    ndi::g_prof_pool[r.dev].push_back(r.b);
  }
  ndi::g_prof_recs.clear();
  ndi::g_prof_acc.last_path = ndi::g_last_path.load();
  *out = ndi::g_prof_acc;
  if (reset) ndi::g_prof_acc = ndi_profile{};
  return NDI_OK;
  NDI_CATCH
}
