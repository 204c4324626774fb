// C ABI of libzkhip.so (include/zkhip.h): contexts, device buffers, registered-base residency, and the host-buffer
// wrappers around the device paths in msm.hip / ntt.hip.  No CPU arithmetic lives here.
//
// Concurrency model (callers are the thread running create_proof and possibly rayon workers):
//   * one context per HIP device named in zkhip_init; devs[0] is the PRIMARY device: every `_device` entry point, every NTT and every
//     single-shard MSM runs there.  The other devices only ever run MSM shards.
//   * scratch memory is keyed by stream.  A `_device` call uses the scratch set of the caller's stream (work on one stream is ordered,
//     so its calls may share scratch; two streams never do).  Host-buffer calls borrow a LANE -- a library-owned stream with its own
//     scratch set and pinned result buffer -- for the duration of the call and do NOT hold the library lock while they wait for the
//     device: two host threads overlap one call's PCIe transfer with the other's kernels.
//   * g_mu guards the context tables (registered bases, handles, scratch map, lanes) and the enqueue of `_device` calls.
//   * an MSM over registered bases that span several shards (zkhip_init with ndev > 1, or ZKHIP_SHARDS / zkhip_set_msm_shards virtual
//     shards on one device) fans the scalar slices out to the shards' devices, gathers the 96-byte Jacobian partials on the primary
//     device (peer copies over xGMI) and folds them there (k_sum_jacobian) -- SURVEY.md section 8(e).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "zkhip_internal.hpp"
#include "../../include/zkhip.hpp"   // host-side 4 x 64 Montgomery arithmetic for domain constants (zkhip::halo2::detail)

namespace zkhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiler (one profiling caller at a time: a debugging aid, not part of the concurrent surface) --------------------
// mode 1: the events of the LAST call are kept (read after every call: a host read-back between calls).
// mode 2: every call appends its events to a pool and zkhip_profile_read_calls reads them all at the end, so a loop of calls runs
//         back to back as it does unprofiled (bench.py's roofline figure; round 4 read back after every call and interleaved other work,
//         which put the phase sum above the step time).
static int g_prof_mode = 0;
static const int PROF_MAX = 32;
static const int PROF_POOL = 2048;           // mode 2: events of up to PROF_POOL / 16 typical calls
static hipEvent_t g_prof_ev[PROF_POOL];
static bool g_prof_ev_ready = false;
static int g_prof_base = 0;                  // first event of the current call
static int g_prof_n = 0;                     // number of marks recorded after the current call's begin event
static char g_prof_names[PROF_POOL][24];
static int g_prof_calls[PROF_POOL / 2][2];   // mode 2: (base, marks) per recorded call
static int g_prof_ncalls = 0;
static bool g_prof_full = false;             // mode 2: the pool ran out, the current call records nothing

void prof_begin(hipStream_t stream) {
  if (!g_prof_mode) return;
  if (!g_prof_ev_ready) {
    for (int i = 0; i < PROF_POOL; i++) (void)hipEventCreate(&g_prof_ev[i]);
    g_prof_ev_ready = true;
  }
  if (g_prof_mode == 2) {
    if (g_prof_full) return;
    int base = 0;
    if (g_prof_ncalls > 0) {                 // close the previous call
      g_prof_calls[g_prof_ncalls - 1][1] = g_prof_n;
      base = g_prof_calls[g_prof_ncalls - 1][0] + g_prof_n + 1;
    }
    if (base + PROF_MAX + 1 > PROF_POOL || g_prof_ncalls >= PROF_POOL / 2) { g_prof_full = true; g_prof_n = PROF_MAX; return; }
    g_prof_base = base;
    g_prof_calls[g_prof_ncalls][0] = base;
    g_prof_calls[g_prof_ncalls][1] = 0;
    g_prof_ncalls++;
  } else {
    g_prof_base = 0;
  }
  g_prof_n = 0;
  (void)hipEventRecord(g_prof_ev[g_prof_base], stream);
}

void prof_mark(hipStream_t stream, const char* name) {
  if (!g_prof_mode || !g_prof_ev_ready || g_prof_n >= PROF_MAX) return;
  snprintf(g_prof_names[g_prof_base + g_prof_n], sizeof(g_prof_names[0]), "%s", name);
  g_prof_n++;
  (void)hipEventRecord(g_prof_ev[g_prof_base + g_prof_n], stream);
}


#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

// roctx ranges, one per C-ABI call, so that `rocprofv3 --marker-trace --kernel-trace` attributes kernels to entry points (the tracing
// counterpart of the reference's per-phase `start_timer!` lines).  OPT-IN: the ranges are live when a marker library is
// ALREADY loaded in the process (a profiler brought it: found with RTLD_NOLOAD, nothing is loaded on the library's own initiative) or when
// ZKHIP_ROCTX=1 asks for it (then rocprofiler-sdk's roctx, else roctracer's, is dlopen'ed); otherwise -- every production host -- the macro
// costs one branch on a cached null pointer.  ZKHIP_NO_ROCTX=1 turns them off under a profiler too.  No link dependency either way.
struct roctx_api {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
static const roctx_api& roctx() {
  static const roctx_api api = [] {
    roctx_api a;
    if (getenv("ZKHIP_NO_ROCTX")) return a;
    static const char* const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void* h = nullptr;
    for (const char* nm : names) if (!h) h = dlopen(nm, RTLD_LAZY | RTLD_LOCAL | RTLD_NOLOAD);
    const char* want = getenv("ZKHIP_ROCTX");
    if (!h && want && want[0] == '1')
      for (const char* nm : names) if (!h) h = dlopen(nm, RTLD_LAZY | RTLD_LOCAL);
    if (h) {
      a.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      a.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    }
    return a;
  }();
  return api;
}
struct api_range {
  bool on;
  explicit api_range(const char* name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
  ~api_range() { if (on) (void)roctx().pop(); }
  api_range(const api_range&) = delete;
  api_range& operator=(const api_range&) = delete;
};
#define ZK_API_RANGE() zkhip::api_range zk_api_range_(__func__)

struct dev_buf {       // grow-only device scratch (hipFree waits for the device, so growing under queued work is safe)
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return ZKHIP_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8;
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu) failed", bytes); p = nullptr; return ZKHIP_ENOMEM; }
      want = bytes;
    }
    cap = want;
    return ZKHIP_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct scratch {       // one user at a time: the calls of one stream
  dev_buf ws, scalars, bases, poly, poly2, small, ntt_tmp, vm, gather;
  vm_staging vm_stage;                 // pinned host staging of the row programs' blobs (rowvm.hip)
  arg_ring args;                       // pinned slots for the small host arrays of `_device` calls (lists of column addresses, coefficients)
  uint64_t last_use = 0;
  void release() { ws.release(); scalars.release(); bases.release(); poly.release(); poly2.release(); small.release(); ntt_tmp.release(); vm.release(); gather.release(); vm_stage.release(); args.release(); }
};

constexpr int STREAM_PIECES_MAX = 64;      // pieces of one chunked host-buffer MSM
struct lane {          // host-buffer calls: a library-owned stream + (through the stream) a scratch set + a pinned result buffer
  hipStream_t stream = nullptr;
  void* pinned = nullptr;              // LANE_PINNED bytes, hipHostMalloc
  bool busy = false;
  // chunked host-buffer MSM (msm_shard_enqueue): the scalar pieces cross PCIe on `copy` while the pieces before them are sorted and
  // accumulated on `stream`; one event per piece in flight.  Created on first use.
  hipStream_t copy = nullptr;
  hipEvent_t copied[STREAM_PIECES_MAX] = {};
};
constexpr size_t LANE_PINNED = 64 * 1024;
constexpr size_t MAX_STREAM_SCRATCH = 8;   // scratch sets kept per device (least recently used caller streams are dropped beyond that)

// a host thread that drives one secondary device during a sharded MSM (its own PCIe link is fed by its own thread: a pageable
// hipMemcpyAsync stages through the calling thread)
struct worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
  char err[512] = "";
  void loop() {
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<int()> j = std::move(job);
      has_job = false;
      lk.unlock();
      const int r = j();
      lk.lock();
      rc = r;
      snprintf(err, sizeof(err), "%s", g_err);     // the worker thread's error text, for the caller
      done = true;
      cv.notify_all();
    }
  }
  void submit(std::function<int()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j); has_job = true; done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    done = false;
    if (rc != ZKHIP_OK) set_error("%s", err);
    return rc;
  }
};

struct device_ctx {
  int device = 0;
  std::vector<lane> lanes;
  std::map<hipStream_t, scratch*> scratch_by_stream;
  uint64_t use_clock = 0;
  dev_buf fixed_table;                 // multiples of the generator for zkhip_g1_fixed_base_mul_device (primary only)
  bool fixed_table_ready = false;
  dev_buf gather;                      // primary: the shards' 96-byte partials, one 96-byte slot per shard
  std::vector<hipEvent_t> shard_events;
  hipStream_t side = nullptr;          // primary: second stream of a batch of large MSMs (high priority: never the caller's hardware queue)
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  worker* w = nullptr;                 // secondary devices only
  // device-resident sharded commits (zkhip_msm_g1_registered_device over several shards): a secondary device runs its shards on `fan`, a
  // stream of its own (not the lane its worker thread drives for host-buffer calls), behind `fan_ready` of the primary device (recorded on
  // the caller's stream: the scalars are complete) and records `fan_done` for the caller's stream to wait on before the fold
  hipStream_t fan = nullptr;
  hipEvent_t fan_ready = nullptr, fan_done = nullptr;
};

struct shard_t {       // points [lo, lo + n) of a registered array, prepared on device `dev`
  int dev = 0;         // index into g_ctx.devs
  size_t lo = 0, n = 0;
  prepared_bases* pb = nullptr;
};

constexpr int FINGER_POINTS = 8;
struct registered_t {  // one zkhip_register_bases call
  const uint64_t* host = nullptr;
  size_t n = 0;
  std::vector<shard_t> shards;
  size_t finger_idx[FINGER_POINTS];
  uint64_t finger[FINGER_POINTS][8];   // sampled points of the host array at registration: a stale or modified range is not trusted
  ~registered_t();
};

struct context {
  bool ready = false;
  std::vector<device_ctx*> devs;       // devs[0] = primary
  int shards = 1;                      // MSM shards a registered array is cut into (default: one per device)
  int ntt_fanout = 1;                  // batched transforms over the devices: 0 never, 1 host-buffer forms, 2 `_device` forms too (zkhip_set_ntt_fanout)
  std::map<const void*, std::shared_ptr<registered_t>> registered;
  std::map<uint64_t, prepared_bases*> handles;          // zkhip_prepare_bases_device handles (primary device)
  uint64_t next_handle = 1;
};

static std::recursive_mutex g_mu;
static std::condition_variable_any g_lane_cv;
static std::mutex g_fanout_mu;         // one multi-device MSM at a time (the secondary devices have one lane each)
static context g_ctx;

static inline device_ctx& primary() { return *g_ctx.devs[0]; }

registered_t::~registered_t() {
  for (auto& sh : shards) {
    if (!sh.pb) continue;
    if (g_ctx.ready && sh.dev < (int)g_ctx.devs.size()) (void)hipSetDevice(g_ctx.devs[sh.dev]->device);
    release_prepared(sh.pb);
  }
  if (g_ctx.ready) (void)hipSetDevice(primary().device);
}

static int ensure_init() {
  if (!g_ctx.ready) return zkhip_init(nullptr, 0);
  // the current device is per host thread: a rayon worker that never touched HIP starts on device 0
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != primary().device) {
    if (hipSetDevice(primary().device) != hipSuccess) { set_error("hipSetDevice(%d) failed", primary().device); return ZKHIP_ENODEV; }
  }
  return ZKHIP_OK;
}

// scratch set of `stream` on device d (g_mu held).  Beyond MAX_STREAM_SCRATCH sets the least recently used one that does not belong
// to a library lane is dropped (after a device synchronisation: its stream may have been destroyed by the caller).
static scratch* scratch_for(device_ctx& d, hipStream_t stream) {
  auto it = d.scratch_by_stream.find(stream);
  if (it != d.scratch_by_stream.end()) { it->second->last_use = ++d.use_clock; return it->second; }
  if (d.scratch_by_stream.size() >= MAX_STREAM_SCRATCH + d.lanes.size()) {
    auto victim = d.scratch_by_stream.end();
    for (auto jt = d.scratch_by_stream.begin(); jt != d.scratch_by_stream.end(); ++jt) {
      bool is_lane = (jt->first == d.side && d.side != nullptr) || (jt->first == d.fan && d.fan != nullptr);
      for (auto& L : d.lanes) is_lane |= (L.stream == jt->first);
      if (is_lane) continue;
      if (victim == d.scratch_by_stream.end() || jt->second->last_use < victim->second->last_use) victim = jt;
    }
    if (victim != d.scratch_by_stream.end()) {
      (void)hipDeviceSynchronize();
      victim->second->release();
      delete victim->second;
      d.scratch_by_stream.erase(victim);
    }
  }
  scratch* sc = new scratch();
  sc->last_use = ++d.use_clock;
  d.scratch_by_stream[stream] = sc;
  return sc;
}

// `stream` argument of the `_device` entry points: NULL is HIP's default (legacy) stream -- stream 0, which is what a caller
// holding "the default stream" (e.g. torch.cuda.current_stream().cuda_stream == 0) passes, and which is ordered with that caller's
// other default-stream work.
static inline hipStream_t caller_stream(void* stream) { return (hipStream_t)stream; }

// ---- lanes -----------------------------------------------------------------------------------------------------------
// RAII: borrows a free lane of the primary device (waits for one), resolves its scratch set.  Construct WITHOUT holding g_mu.
struct lane_hold {
  lane* L = nullptr;
  scratch* sc = nullptr;
  hipStream_t s = nullptr;
  int rc = ZKHIP_OK;
  lane_hold() {
    std::unique_lock<std::recursive_mutex> lk(g_mu);
    rc = ensure_init();
    if (rc != ZKHIP_OK) return;
    for (;;) {
      for (auto& c : primary().lanes) if (!c.busy) { L = &c; break; }
      if (L) break;
      g_lane_cv.wait(lk);
      if (!g_ctx.ready) { rc = ZKHIP_ENODEV; set_error("library shut down while a call was waiting"); return; }
    }
    L->busy = true;
    s = L->stream;
    sc = scratch_for(primary(), s);
  }
  ~lane_hold() {
    if (!L) return;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    L->busy = false;
    g_lane_cv.notify_one();
  }
  lane_hold(const lane_hold&) = delete;
  lane_hold& operator=(const lane_hold&) = delete;
};

// Contiguous point range of shard s of S over n points: the first n % S shards get one extra point (the partition the multi-process
// path uses too: zksnap_circuits_halo2_amd/multi_gpu.py shard_range)
static inline void shard_range(size_t n, int s, int S, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)S, extra = n % (size_t)S;
  *lo = (size_t)s * base + std::min<size_t>((size_t)s, extra);
  *hi = *lo + base + ((size_t)s < extra ? 1 : 0);
}

// registered entry + point offset for `bases` if it lies inside a registered range with room for n points and the sampled points
// of that range still hold what was registered (g_mu held)
static std::shared_ptr<registered_t> find_registered(const uint64_t* bases, size_t n, size_t* off) {
  for (auto& kv : g_ctx.registered) {
    const char* lo = (const char*)kv.first;
    const char* hi = lo + kv.second->n * 64;
    const char* q = (const char*)bases;
    if (q >= lo && q + n * 64 <= hi && ((q - lo) % 64) == 0) {
      const registered_t& r = *kv.second;
      for (int i = 0; i < FINGER_POINTS; i++)
        if (memcmp(r.host + r.finger_idx[i] * 8, r.finger[i], 64) != 0) return nullptr;   // reused or modified memory: not the registered SRS
      *off = (size_t)(q - lo) / 64;
      return kv.second;
    }
  }
  return nullptr;
}

static void destroy_device_ctx(device_ctx* d) {
  (void)hipSetDevice(d->device);
  (void)hipDeviceSynchronize();
  if (d->w) {
    { std::lock_guard<std::mutex> lk(d->w->mu); d->w->quit = true; d->w->cv.notify_all(); }
    if (d->w->th.joinable()) d->w->th.join();
    delete d->w;
  }
  for (auto& kv : d->scratch_by_stream) { kv.second->release(); delete kv.second; }
  d->scratch_by_stream.clear();
  for (auto& L : d->lanes) {
    if (L.pinned) (void)hipHostFree(L.pinned);
    if (L.stream) (void)hipStreamDestroy(L.stream);
    if (L.copy) (void)hipStreamDestroy(L.copy);
    for (auto& e : L.copied) if (e) (void)hipEventDestroy(e);
  }
  for (auto e : d->shard_events) (void)hipEventDestroy(e);
  if (d->side_fork) (void)hipEventDestroy(d->side_fork);
  if (d->side_join) (void)hipEventDestroy(d->side_join);
  if (d->side) (void)hipStreamDestroy(d->side);
  if (d->fan_ready) (void)hipEventDestroy(d->fan_ready);
  if (d->fan_done) (void)hipEventDestroy(d->fan_done);
  if (d->fan) (void)hipStreamDestroy(d->fan);
  d->fixed_table.release();
  d->gather.release();
  delete d;
}

}  // namespace zkhip

using namespace zkhip;
typedef std::lock_guard<std::recursive_mutex> guard_t;

extern "C" {

int zkhip_init(const int* devices, int ndev) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  if (ndev < 0 || ndev > 64) { set_error("zkhip_init: ndev = %d out of range", ndev); return ZKHIP_EINVAL; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); set_error("no HIP device available"); return ZKHIP_ENODEV; }
  std::vector<int> want;
  if (devices && ndev >= 1) want.assign(devices, devices + ndev);
  else if (const char* e = getenv("ZKHIP_DEVICE")) want.push_back(atoi(e));
  else want.push_back(0);
  for (size_t i = 0; i < want.size(); i++) {
    if (want[i] < 0 || want[i] >= count) { set_error("device %d out of range (%d devices)", want[i], count); return ZKHIP_EINVAL; }
    // a device named twice is an error -- except under ZKHIP_TEST_DUPLICATE_DEVICES=1, which lets a one-GPU box rehearse the
    // multi-device machinery (worker threads, peer copies, gather + fold) with several contexts on the same card
    const bool dup_ok = getenv("ZKHIP_TEST_DUPLICATE_DEVICES") != nullptr;
    for (size_t j = 0; j < i && !dup_ok; j++) if (want[j] == want[i]) { set_error("zkhip_init: device %d named twice", want[i]); return ZKHIP_EINVAL; }
  }
  if (g_ctx.ready) {
    bool same = g_ctx.devs.size() == want.size();
    for (size_t i = 0; same && i < want.size(); i++) same = g_ctx.devs[i]->device == want[i];
    if (same) return ensure_init();
    // a host-buffer call in flight holds a lane without the lock; shutdown would wait for it on a mutex this thread holds twice
    // (condition_variable_any releases one level only): refuse instead of deadlocking
    for (auto& L : primary().lanes) if (L.busy) { set_error("zkhip_init: a different device list while host-buffer calls are in flight"); return ZKHIP_EBUSY; }
    zkhip_shutdown();
  }
  int lanes = 2;                                            // host-buffer calls that may be in flight at once
  if (const char* e = getenv("ZKHIP_HOST_LANES")) { const int v = atoi(e); if (v >= 1 && v <= 4) lanes = v; }
  for (size_t i = 0; i < want.size(); i++) {
    if (hipSetDevice(want[i]) != hipSuccess) { (void)hipGetLastError(); set_error("hipSetDevice(%d) failed", want[i]); for (auto d : g_ctx.devs) destroy_device_ctx(d); g_ctx.devs.clear(); return ZKHIP_ENODEV; }
    device_ctx* d = new device_ctx();
    d->device = want[i];
    d->lanes.resize(i == 0 ? (size_t)lanes : 1);
    bool ok = true;
    for (auto& L : d->lanes) {
      ok = ok && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipHostMalloc(&L.pinned, LANE_PINNED, hipHostMallocDefault) == hipSuccess;
    }
    g_ctx.devs.push_back(d);
    if (!ok) { (void)hipGetLastError(); set_error("zkhip_init: stream / pinned buffer creation failed on device %d", want[i]); for (auto dd : g_ctx.devs) destroy_device_ctx(dd); g_ctx.devs.clear(); return ZKHIP_ENODEV; }
    if (i > 0) {
      d->w = new worker();
      const int devno = want[i];
      worker* w = d->w;
      w->th = std::thread([w, devno] { (void)hipSetDevice(devno); w->loop(); });
    }
  }
  // partial sums travel device to device (xGMI); without peer access hipMemcpyPeerAsync stages through the host, which is also fine
  for (size_t i = 1; i < want.size(); i++) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, want[i], want[0]) == hipSuccess && can) {
      (void)hipSetDevice(want[i]);
      (void)hipDeviceEnablePeerAccess(want[0], 0);
    }
    (void)hipGetLastError();
  }
  (void)hipSetDevice(want[0]);
  g_ctx.shards = (int)want.size();
  if (const char* e = getenv("ZKHIP_SHARDS")) { const int v = atoi(e); if (v >= 1 && v <= 64) g_ctx.shards = v; }
  g_ctx.ntt_fanout = 1;
  if (const char* e = getenv("ZKHIP_NTT_FANOUT")) { const int v = atoi(e); if (v >= 0 && v <= 2) g_ctx.ntt_fanout = v; }
  g_ctx.ready = true;
  return ZKHIP_OK;
}

void zkhip_shutdown(void) {
  ZK_API_RANGE();
  std::unique_lock<std::recursive_mutex> lk(g_mu);
  if (!g_ctx.ready) return;
  // host-buffer calls in flight hold a lane without the lock: let them finish
  for (;;) {
    bool busy = false;
    for (auto& L : primary().lanes) busy |= L.busy;
    if (!busy) break;
    g_lane_cv.wait(lk);
  }
  (void)hipSetDevice(primary().device);
  (void)hipDeviceSynchronize();
  ntt_clear_cache();
  row_vm_jit_clear();
  g_ctx.registered.clear();                                  // last references: tables are freed by ~registered_t
  for (auto& kv : g_ctx.handles) release_prepared(kv.second);
  g_ctx.handles.clear();
  for (auto d : g_ctx.devs) destroy_device_ctx(d);
  g_ctx.devs.clear();
  g_ctx.ready = false;
  g_lane_cv.notify_all();
}

const char* zkhip_last_error(void) { return g_err; }

int zkhip_device_name(char* buf, size_t len) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, primary().device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return ZKHIP_OK;
}

int zkhip_device_count(void) {
  guard_t g(g_mu);
  if (ensure_init() != ZKHIP_OK) return 0;
  return (int)g_ctx.devs.size();
}

int zkhip_set_msm_shards(int shards) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (shards < 0 || shards > 64) { set_error("set_msm_shards: %d out of range [0, 64]", shards); return ZKHIP_EINVAL; }
  g_ctx.shards = shards == 0 ? (int)g_ctx.devs.size() : shards;
  return ZKHIP_OK;
}

int zkhip_msm_shards(void) {
  guard_t g(g_mu);
  if (ensure_init() != ZKHIP_OK) return 0;
  return g_ctx.shards;
}

int zkhip_msm_window_bits(size_t n) { return msm_pick_window(n); }

// ---- MSM -----------------------------------------------------------------------------------------
int zkhip_msm_g1_device_c(const void* d_scalars, const void* d_bases, size_t n, void* d_out_xyz, int window_bits, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_out_xyz || (n && (!d_scalars || !d_bases))) { set_error("msm: null pointer"); return ZKHIP_EINVAL; }
  const int c = window_bits > 0 ? window_bits : msm_pick_window(n ? n : 1);
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(msm_workspace_bytes(n, c))) != ZKHIP_OK) return rc;
  return msm_g1_device((const uint32_t*)d_scalars, (const uint32_t*)d_bases, n, (uint32_t*)d_out_xyz, sc->ws.p, sc->ws.cap, c, s);
}

int zkhip_msm_g1_device(const void* d_scalars, const void* d_bases, size_t n, void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  return zkhip_msm_g1_device_c(d_scalars, d_bases, n, d_out_xyz, 0, stream);
}

}  // extern "C"

namespace zkhip {

// One shard's share of a host-buffer MSM on device `d` (current device = d.device, stream / scratch of its lane 0 or the borrowed
// lane): uploads the scalar slice (and the base slice when there is no prepared table), runs the MSM, leaves the 96-byte partial at
// d_partial (memory of device d).
// Chunked upload (round 4).  A host-buffer MSM used to upload all its scalars with one copy and only then start its first kernel (2^22 points:
// 2.4 ms of PCIe + 4.9 ms of kernels, nothing overlapped).  With a prepared table of wide windows and at least two pieces of 2^20 scalars the
// upload is cut into J even pieces on the lane's copy stream; piece j is sorted and accumulated into the shared bucket set (msm.hip:
// msm_chunk_add) while piece j + 1 crosses PCIe, and one reduction tail ends the call.  Only the first piece's upload is exposed.
// ZKHIP_STREAM_PIECE_LOG: log2 of the target piece size (default 20; 0 = never chunk).
static size_t msm_stream_pieces(size_t n, const prepared_bases* pb) {
  static const int piece_log = [] { const char* e = getenv("ZKHIP_STREAM_PIECE_LOG"); const int v = e ? atoi(e) : 20; return (v == 0 || (v >= 16 && v <= 26)) ? v : 20; }();
  if (!pb || pb->c <= 16 || piece_log == 0) return 1;
  const size_t J = n >> piece_log;
  return J < 2 ? 1 : std::min<size_t>(J, (size_t)STREAM_PIECES_MAX);
}

static int msm_shard_enqueue_chunked(lane* L, scratch* sc, hipStream_t s, const uint64_t* scalars, size_t n, const prepared_bases* pb, size_t pb_off,
                                     uint32_t* d_partial, size_t J, size_t region) {
  int rc;
  if (!L->copy) HIPCHK(hipStreamCreateWithFlags(&L->copy, hipStreamNonBlocking));
  const size_t cap = (n + J - 1) / J;
  const size_t ws_bytes = msm_chunk_workspace_bytes(cap, pb->c);
  if ((rc = sc->scalars.reserve((region + n) * 32)) != ZKHIP_OK) return rc;      // (a no-op inside a fan-out: reserve_for_pieces sized the buffer for all pieces)
  if ((rc = sc->ws.reserve(ws_bytes)) != ZKHIP_OK) return rc;
  // (round 4 advice) every non-OK exit below waits for the copy stream and for `s`: the caller's scalars may be pinned host memory, in which case an
  // earlier piece could still be crossing PCIe from the caller's buffer into the lane's after this call has returned its error and released the lane
  struct drain_t {
    hipStream_t a, b;
    bool armed = true;
    ~drain_t() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); } }
  } drain{L->copy, s};
  // Every piece of a call has its own region of the lane's scalar buffer (`region`, in scalars), so the copy stream may run ahead of the kernels:
  // while shard k is being accumulated on `s`, the pieces of shard k + 1 are already crossing PCIe (virtual shards on one lane).  A call ends with
  // hipStreamSynchronize(s), and `s` waits for every copy it used, so nothing of an earlier call is in flight here.
  char* const dst = (char*)sc->scalars.p + region * 32;
  size_t lo = 0;
  for (size_t j = 0; j < J; j++) {
    const size_t len = n / J + (j < n % J ? 1 : 0);          // cap or cap - 1
    if (!L->copied[j]) HIPCHK(hipEventCreateWithFlags(&L->copied[j], hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(dst + lo * 32, scalars + lo * 4, len * 32, hipMemcpyHostToDevice, L->copy));
    HIPCHK(hipEventRecord(L->copied[j], L->copy));
    HIPCHK(hipStreamWaitEvent(s, L->copied[j], 0));
    if ((rc = msm_chunk_add((const uint32_t*)(dst + lo * 32), len, pb, pb_off + lo, cap, j == 0, sc->ws.p, sc->ws.cap, s)) != ZKHIP_OK) return rc;
    lo += len;
  }
  if ((rc = msm_chunk_finish(pb, cap, d_partial, sc->ws.p, sc->ws.cap, s)) != ZKHIP_OK) return rc;
  drain.armed = false;
  return ZKHIP_OK;
}

// region: offset (in scalars) of this piece inside the lane's scalar buffer -- the pieces one lane runs back to back (virtual shards) do not share
// upload space, so one piece's upload never waits for the previous piece's kernels
static int msm_shard_enqueue(scratch* sc, hipStream_t s, const uint64_t* scalars, const uint64_t* bases, size_t n, const prepared_bases* pb,
                             size_t pb_off, uint32_t* d_partial, lane* L = nullptr, size_t region = 0) {
  int rc;
  if (n == 0) return msm_g1_device(nullptr, nullptr, 0, d_partial, nullptr, 0, 0, s);
  if (L) {
    const size_t J = msm_stream_pieces(n, pb);
    if (J > 1) return msm_shard_enqueue_chunked(L, sc, s, scalars, n, pb, pb_off, d_partial, J, region);
  }
  if ((rc = sc->scalars.reserve((region + n) * 32)) != ZKHIP_OK) return rc;
  char* const d_sc = (char*)sc->scalars.p + region * 32;
  HIPCHK(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, s));
  if (pb) {
    if ((rc = sc->ws.reserve(msm_workspace_bytes(n, pb->c, true))) != ZKHIP_OK) return rc;
    return msm_g1_device((const uint32_t*)d_sc, nullptr, n, d_partial, sc->ws.p, sc->ws.cap, 0, s, pb, pb_off);
  }
  if ((rc = sc->bases.reserve(n * 64)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(sc->bases.p, bases, n * 64, hipMemcpyHostToDevice, s));
  const int c = msm_pick_window(n);
  if ((rc = sc->ws.reserve(msm_workspace_bytes(n, c))) != ZKHIP_OK) return rc;
  return msm_g1_device((const uint32_t*)d_sc, (const uint32_t*)sc->bases.p, n, d_partial, sc->ws.p, sc->ws.cap, c, s);
}

struct piece_t { int dev; size_t lo, n; const prepared_bases* pb; size_t pb_off; };   // points [lo, lo + n) of the call's range

// Size the scratch for the pieces one device will run back to back (scalars: the sum, one region per piece; bases and workspace: the largest):
// growing a buffer between two pieces would free it (a device-wide wait) under the previous piece's kernels.
static int reserve_for_pieces(scratch* sc, const std::vector<piece_t>& pieces, const std::vector<size_t>& mine) {
  size_t sc_bytes = 0, bs_bytes = 0, ws_bytes = 0;
  for (size_t i : mine) {
    const piece_t& p = pieces[i];
    if (p.n == 0) continue;
    sc_bytes += p.n * 32;                           // one region per piece (msm_shard_enqueue)
    if (!p.pb) bs_bytes = std::max(bs_bytes, p.n * 64);
    const size_t J = msm_stream_pieces(p.n, p.pb);
    if (J > 1) ws_bytes = std::max(ws_bytes, msm_chunk_workspace_bytes((p.n + J - 1) / J, p.pb->c));
    else ws_bytes = std::max(ws_bytes, p.pb ? msm_workspace_bytes(p.n, p.pb->c, true) : msm_workspace_bytes(p.n, msm_pick_window(p.n)));
  }
  int rc;
  if ((rc = sc->scalars.reserve(sc_bytes)) != ZKHIP_OK) return rc;
  if ((rc = sc->bases.reserve(bs_bytes)) != ZKHIP_OK) return rc;
  return sc->ws.reserve(ws_bytes);
}

// Host-buffer MSM on a borrowed lane: out = sum scalars[i] * bases[i].  `reg` (may be null) = the registered entry `bases` points
// into at point offset `off`.  Pieces = the shards the range touches (registered) or an even split over the devices (not registered,
// several devices); a single piece on the primary device is the plain path.
static int host_msm(lane_hold& H, const uint64_t* scalars, const uint64_t* bases, size_t n, const std::shared_ptr<registered_t>& reg, size_t off,
                    uint64_t* out_xyz) {
  int rc;
  std::vector<piece_t> pieces;
  if (reg) {
    for (auto& sh : reg->shards) {
      const size_t lo = std::max(off, sh.lo), hi = std::min(off + n, sh.lo + sh.n);
      if (lo < hi) pieces.push_back({sh.dev, lo - off, hi - lo, sh.pb, lo - sh.lo});
    }
  } else if (g_ctx.devs.size() > 1 && n >= 4096 * g_ctx.devs.size()) {
    const int S = (int)g_ctx.devs.size();
    for (int s = 0; s < S; s++) {
      size_t lo, hi;
      shard_range(n, s, S, &lo, &hi);
      if (lo < hi) pieces.push_back({s, lo, hi - lo, nullptr, 0});
    }
  }
  if (pieces.empty()) pieces.push_back({0, 0, n, nullptr, 0});
  hipStream_t s = H.s;
  scratch* sc = H.sc;
  if ((rc = sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  if (pieces.size() == 1 && pieces[0].dev == 0) {
    const piece_t& p = pieces[0];
    if ((rc = msm_shard_enqueue(sc, s, scalars + p.lo * 4, bases + p.lo * 8, p.n, p.pb, p.pb_off, (uint32_t*)sc->small.p, H.L)) != ZKHIP_OK) return rc;
  } else {
    // fan out: the primary device's pieces run on this thread's lane, one after another (virtual shards) -- the other devices'
    // pieces on their worker threads, each device over its own PCIe link
    device_ctx& P = primary();
    std::unique_lock<std::mutex> fan(g_fanout_mu, std::defer_lock);
    bool remote = false;
    for (auto& p : pieces) remote |= p.dev != 0;
    if (remote) fan.lock();
    // gather slots live in the lane's scratch set: concurrent single-device fan-outs on different lanes do not share them
    if ((rc = sc->poly2.reserve(pieces.size() * 96 + 256)) != ZKHIP_OK) return rc;
    uint32_t* gather = (uint32_t*)sc->poly2.p;
    std::vector<std::vector<size_t>> by_dev(g_ctx.devs.size());
    for (size_t i = 0; i < pieces.size(); i++) by_dev[(size_t)pieces[i].dev].push_back(i);
    if ((rc = reserve_for_pieces(sc, pieces, by_dev[0])) != ZKHIP_OK) return rc;
    for (size_t d = 1; d < g_ctx.devs.size(); d++) {
      if (by_dev[d].empty()) continue;
      device_ctx* D = g_ctx.devs[d];
      const std::vector<size_t> mine = by_dev[d];
      const int primary_dev = P.device;
      D->w->submit([D, mine, &pieces, scalars, bases, gather, primary_dev]() -> int {
        hipStream_t ds = D->lanes[0].stream;
        scratch* dsc;
        { guard_t g(g_mu); dsc = scratch_for(*D, ds); }
        int r;
        if ((r = dsc->small.reserve(4096 + mine.size() * 96)) != ZKHIP_OK) return r;
        if ((r = reserve_for_pieces(dsc, pieces, mine)) != ZKHIP_OK) return r;
        size_t region = 0;
        for (size_t k = 0; k < mine.size(); k++) {
          const piece_t& p = pieces[mine[k]];
          uint32_t* part = (uint32_t*)((char*)dsc->small.p + 4096 + k * 96);
          if ((r = msm_shard_enqueue(dsc, ds, scalars + p.lo * 4, bases + p.lo * 8, p.n, p.pb, p.pb_off, part, &D->lanes[0], region)) != ZKHIP_OK) return r;
          region += p.n;
          HIPCHK(hipMemcpyPeerAsync(gather + mine[k] * 24, primary_dev, part, D->device, 96, ds));
        }
        HIPCHK(hipStreamSynchronize(ds));                 // the partials have landed on the primary device
        return ZKHIP_OK;
      });
    }
    int rc_local = ZKHIP_OK;
    size_t region0 = 0;
    for (size_t i : by_dev[0]) {
      const piece_t& p = pieces[i];
      if ((rc_local = msm_shard_enqueue(sc, s, scalars + p.lo * 4, bases + p.lo * 8, p.n, p.pb, p.pb_off, gather + i * 24, H.L, region0)) != ZKHIP_OK) break;
      region0 += p.n;
    }
    int rc_remote = ZKHIP_OK;
    for (size_t d = 1; d < g_ctx.devs.size(); d++) {
      if (by_dev[d].empty()) continue;
      const int r = g_ctx.devs[d]->w->wait();
      if (r != ZKHIP_OK) rc_remote = r;
    }
    if (rc_local != ZKHIP_OK) return rc_local;
    if (rc_remote != ZKHIP_OK) return rc_remote;
    // the remote partials were written by other devices' streams, which this thread has waited for; the fold is ordered behind
    // the local pieces on the lane's stream
    if ((rc = sum_jacobian_device(gather, (int)pieces.size(), (uint32_t*)sc->small.p, s)) != ZKHIP_OK) return rc;
  }
  HIPCHK(hipMemcpyAsync(H.L->pinned, sc->small.p, 96, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  memcpy(out_xyz, H.L->pinned, 96);
  return ZKHIP_OK;
}

}  // namespace zkhip

extern "C" {

int zkhip_msm_g1(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xyz[12]) {
  ZK_API_RANGE();
  if (!out_xyz || (n && (!scalars || !bases))) { set_error("msm: null pointer"); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  std::shared_ptr<registered_t> reg;
  size_t off = 0;
  if (n) { guard_t g(g_mu); reg = find_registered(bases, n, &off); }
  return host_msm(H, scalars, bases, n, reg, off, out_xyz);
}

// `batch` scalar vectors (contiguous, n elements each) against the same bases; out: batch Jacobian points.
// Registered bases in one shard: one batched launch set (window <= 16 bits) or pairs of overlapping MSMs (wide windows); otherwise one
// MSM per vector.
int zkhip_msm_g1_batch(const uint64_t* scalars, const uint64_t* bases, size_t n, size_t batch, uint64_t* out_xyz) {
  ZK_API_RANGE();
  if (!out_xyz || (n && batch && (!scalars || !bases))) { set_error("msm_batch: null pointer"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  std::shared_ptr<registered_t> reg;
  size_t off = 0;
  if (n) { guard_t g(g_mu); reg = find_registered(bases, n, &off); }
  const shard_t* one = nullptr;
  if (reg) for (auto& sh : reg->shards) if (sh.dev == 0 && off >= sh.lo && off + n <= sh.lo + sh.n) one = &sh;
  if (!one || batch * 96 > LANE_PINNED || (one->pb->c > 16 && n * batch * 32 > ((size_t)4 << 30))) {   // (wide windows: the device path overlaps pairs of MSMs)
    for (size_t k = 0; k < batch; k++)
      if ((rc = host_msm(H, scalars + k * n * 4, bases, n, reg, off, out_xyz + k * 12)) != ZKHIP_OK) return rc;
    return ZKHIP_OK;
  }
  hipStream_t s = H.s;
  scratch* sc = H.sc;
  if ((rc = sc->scalars.reserve(n * batch * 32)) != ZKHIP_OK) return rc;
  if ((rc = sc->small.reserve(4096 + batch * 96)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(sc->scalars.p, scalars, n * batch * 32, hipMemcpyHostToDevice, s));
  {
    // reuse the handle-based entry point through a temporary handle for the registered table
    guard_t g(g_mu);
    const uint64_t tmp_handle = g_ctx.next_handle++;
    g_ctx.handles[tmp_handle] = one->pb;
    rc = zkhip_msm_g1_prepared_batch_device(tmp_handle, off - one->lo, sc->scalars.p, n, batch, n, sc->small.p, s);
    g_ctx.handles.erase(tmp_handle);
  }
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(H.L->pinned, sc->small.p, batch * 96, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  memcpy(out_xyz, H.L->pinned, batch * 96);
  return ZKHIP_OK;
}

int zkhip_register_bases(const uint64_t* bases, size_t n) {
  ZK_API_RANGE();
  if (!bases || n == 0) { set_error("register_bases: empty"); return ZKHIP_EINVAL; }
  {
    guard_t g(g_mu);
    int rc = ensure_init();
    if (rc != ZKHIP_OK) return rc;
    if (g_ctx.registered.count(bases)) zkhip_unregister_bases(bases);
  }
  lane_hold H;                                        // the primary device's uploads run on a borrowed lane
  if (H.rc != ZKHIP_OK) return H.rc;
  int S, ndev;
  { guard_t g(g_mu); S = g_ctx.shards; ndev = (int)g_ctx.devs.size(); }
  if ((size_t)S > n) S = (int)n;
  // shards on the other devices are uploaded through those devices' single lane, which a multi-device MSM of another thread drives
  // through its workers: the two take turns
  std::unique_lock<std::mutex> fan(g_fanout_mu, std::defer_lock);
  if (ndev > 1 && S > 1) fan.lock();
  auto reg = std::make_shared<registered_t>();
  reg->host = bases;
  reg->n = n;
  for (int i = 0; i < FINGER_POINTS; i++) {
    reg->finger_idx[i] = (size_t)((unsigned __int128)(n - 1) * (unsigned)i / (FINGER_POINTS - 1));
    memcpy(reg->finger[i], bases + reg->finger_idx[i] * 8, 64);
  }
  int rc = ZKHIP_OK;
  for (int s = 0; s < S && rc == ZKHIP_OK; s++) {
    size_t lo, hi;
    shard_range(n, s, S, &lo, &hi);
    if (lo >= hi) continue;
    const int di = s % ndev;
    device_ctx* D = g_ctx.devs[(size_t)di];
    hipStream_t ds = di == 0 ? H.s : D->lanes[0].stream;
    scratch* dsc;
    { guard_t g(g_mu); dsc = di == 0 ? H.sc : scratch_for(*D, ds); }
    if (hipSetDevice(D->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", D->device); rc = ZKHIP_ENODEV; break; }
    shard_t sh;
    sh.dev = di; sh.lo = lo; sh.n = hi - lo;
    if ((rc = dsc->bases.reserve(sh.n * 64)) == ZKHIP_OK) {
      if (hipMemcpyAsync(dsc->bases.p, bases + lo * 8, sh.n * 64, hipMemcpyHostToDevice, ds) != hipSuccess) { set_error("register_bases: upload failed"); rc = ZKHIP_EHIP; }
      else rc = prepare_bases_device((const uint32_t*)dsc->bases.p, sh.n, ds, &sh.pb, 0, /* direct table only for arrays that are small as a whole: */ n);
    }
    if (rc == ZKHIP_OK) reg->shards.push_back(sh);
  }
  (void)hipSetDevice(primary().device);
  if (rc != ZKHIP_OK) return rc;                      // ~registered_t frees the shards built so far
  guard_t g(g_mu);
  g_ctx.registered[bases] = reg;
  return ZKHIP_OK;
}

int zkhip_unregister_bases(const uint64_t* bases) {
  ZK_API_RANGE();
  std::shared_ptr<registered_t> reg;
  {
    guard_t g(g_mu);
    auto it = g_ctx.registered.find(bases);
    if (it == g_ctx.registered.end()) { set_error("unregister_bases: pointer not registered"); return ZKHIP_EINVAL; }
    reg = it->second;
    g_ctx.registered.erase(it);
    for (auto d : g_ctx.devs) { (void)hipSetDevice(d->device); (void)hipDeviceSynchronize(); }   // the tables may still be in use on a stream
    (void)hipSetDevice(primary().device);
  }
  return ZKHIP_OK;                                    // a host call still running holds its own reference; the last one frees the tables
}

int zkhip_prepare_bases_device(const void* d_bases, size_t n, uint64_t* handle) {
  ZK_API_RANGE();
  return zkhip_prepare_bases_device_c(d_bases, n, 0, handle);
}

int zkhip_prepare_bases_device_c(const void* d_bases, size_t n, int window_bits, uint64_t* handle) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_bases || !handle || n == 0) { set_error("prepare_bases: bad argument"); return ZKHIP_EINVAL; }
  prepared_bases* pb = nullptr;
  HIPCHK(hipDeviceSynchronize());           // d_bases may still be being written on the caller's stream; one-time call
  if ((rc = prepare_bases_device((const uint32_t*)d_bases, n, nullptr, &pb, window_bits)) != ZKHIP_OK) return rc;
  *handle = g_ctx.next_handle++;
  g_ctx.handles[*handle] = pb;
  return ZKHIP_OK;
}

int zkhip_release_bases(uint64_t handle) {
  guard_t g(g_mu);
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("release_bases: unknown handle"); return ZKHIP_EINVAL; }
  (void)hipDeviceSynchronize();
  release_prepared(it->second);
  g_ctx.handles.erase(it);
  return ZKHIP_OK;
}

int zkhip_prepared_window_bits(uint64_t handle) {
  guard_t g(g_mu);
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("prepared_window_bits: unknown handle"); return ZKHIP_EINVAL; }
  return it->second->c;
}

int zkhip_msm_g1_prepared_device(uint64_t handle, size_t offset, const void* d_scalars, size_t n, void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("msm_prepared: unknown handle"); return ZKHIP_EINVAL; }
  if (!d_out_xyz || (n && !d_scalars)) { set_error("msm_prepared: null pointer"); return ZKHIP_EINVAL; }
  const prepared_bases* pb = it->second;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(msm_workspace_bytes(n, pb->c, true))) != ZKHIP_OK) return rc;
  return msm_g1_device((const uint32_t*)d_scalars, nullptr, n, (uint32_t*)d_out_xyz, sc->ws.p, sc->ws.cap, 0, s, pb, offset);
}

}  // extern "C"

namespace zkhip {

// `batch` prepared MSMs of n points each on one device (current device = pb's), results at d_out + k * 24 words: tables with windows of
// <= 16 bits share launch sets (groups of vectors, each with its own bucket set), wider ones run one vector after the other.
// vectors of a batch that share one launch set: all of them for windows <= 16 bits, within 2^31 sorted entries and 4 Mi buckets
static size_t batch_group(const prepared_bases* pb, size_t n, size_t batch) {
  size_t group = (pb->c > 16 || batch <= 1) ? 1 : std::min<size_t>(batch, 65535);
  while (group > 1 && ((size_t)((256 + pb->c - 1) / pb->c) * n * group >= (1ull << 31) || (group << (pb->c - 1)) > (1ull << 22))) group = (group + 1) / 2;
  return group;
}

static int prepared_batch_enqueue(scratch* sc, hipStream_t s, const prepared_bases* pb, size_t pb_off, const uint32_t* d_scalars, size_t n, size_t batch,
                                  size_t scalar_stride, uint32_t* d_out) {
  int rc;
  const size_t group = batch_group(pb, n, batch);
  if ((rc = sc->ws.reserve(msm_workspace_bytes(n, pb->c, true, group))) != ZKHIP_OK) return rc;
  for (size_t k0 = 0; k0 < batch; k0 += group) {
    const size_t kk = batch - k0 < group ? batch - k0 : group;
    rc = msm_g1_device(d_scalars + k0 * scalar_stride * 8, nullptr, n, d_out + k0 * 24, sc->ws.p, sc->ws.cap, 0, s, pb, pb_off, kk, scalar_stride);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

// Device-resident scalars (primary device, caller's stream s) against a registered array whose range [off, off + n) spans several shards
// (SURVEY.md section 8(e) partitioning, with the scalars already in HBM): every shard the range touches runs its own prepared Pippenger on
// its device.  The primary device's pieces run on s.  A secondary device waits for the scalars (event on s), pulls its slice of every
// vector from the primary's HBM over xGMI (hipMemcpyPeerAsync: 32 B x n / S per vector -- 64 MiB per GPU for configs[4], about a
// millisecond beside a 2.9 ms shard MSM), runs its pieces on its own stream and sends each 96-byte partial back into the gather buffer of
// the caller's stream; s waits for the secondaries' events and folds: out[k] = sum over pieces of partial[piece][k].  Asynchronous like
// every `_device` call: nothing here waits for a device.  g_mu held by the caller.
static int registered_device_msm(const std::shared_ptr<registered_t>& reg, size_t off, const uint32_t* d_scalars, size_t n, size_t batch, size_t stride,
                                 uint32_t* d_out, hipStream_t s) {
  int rc;
  device_ctx& P = primary();
  scratch* sc = scratch_for(P, s);
  std::vector<piece_t> pieces;
  for (auto& sh : reg->shards) {
    const size_t lo = std::max(off, sh.lo), hi = std::min(off + n, sh.lo + sh.n);
    if (lo < hi) pieces.push_back({sh.dev, lo - off, hi - lo, sh.pb, lo - sh.lo});
  }
  if (pieces.size() == 1 && pieces[0].dev == 0)
    return prepared_batch_enqueue(sc, s, pieces[0].pb, pieces[0].pb_off, d_scalars, n, batch, stride, d_out);
  const size_t np = pieces.size();
  if ((rc = sc->gather.reserve(np * batch * 96)) != ZKHIP_OK) return rc;
  uint32_t* gather = (uint32_t*)sc->gather.p;                     // [piece][vector] 96-byte slots
  bool remote = false;
  for (auto& p : pieces) remote |= p.dev != 0;
  if (remote) {
    if (!P.fan_ready) HIPCHK(hipEventCreateWithFlags(&P.fan_ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(P.fan_ready, s));
  }
  // the secondaries first: their copies and kernels start while this thread is still enqueueing the primary's pieces
  int rc_remote = ZKHIP_OK;
  std::vector<size_t> waited;
  for (size_t d = 1; d < g_ctx.devs.size() && rc_remote == ZKHIP_OK; d++) {
    std::vector<size_t> mine;
    for (size_t i = 0; i < np; i++) if ((size_t)pieces[i].dev == d) mine.push_back(i);
    if (mine.empty()) continue;
    device_ctx* D = g_ctx.devs[d];
    if (hipSetDevice(D->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", D->device); rc_remote = ZKHIP_ENODEV; break; }
    auto body = [&]() -> int {
      if (!D->fan) {
        HIPCHK(hipStreamCreateWithFlags(&D->fan, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&D->fan_done, hipEventDisableTiming));
      }
      scratch* dsc = scratch_for(*D, D->fan);
      size_t sc_elems = 0;
      for (size_t i : mine) sc_elems = std::max(sc_elems, pieces[i].n);
      int r;
      if ((r = dsc->scalars.reserve(sc_elems * batch * 32)) != ZKHIP_OK) return r;
      if ((r = dsc->small.reserve(4096 + batch * 96)) != ZKHIP_OK) return r;
      size_t ws_bytes = 0;                                        // sized once for the largest piece (growing it later would free it under queued work)
      for (size_t i : mine) ws_bytes = std::max(ws_bytes, msm_workspace_bytes(pieces[i].n, pieces[i].pb->c, true, batch_group(pieces[i].pb, pieces[i].n, batch)));
      if ((r = dsc->ws.reserve(ws_bytes)) != ZKHIP_OK) return r;
      HIPCHK(hipStreamWaitEvent(D->fan, P.fan_ready, 0));
      for (size_t i : mine) {
        const piece_t& p = pieces[i];
        uint32_t* slice = (uint32_t*)dsc->scalars.p;              // vector k of this piece at slice + k * p.n * 8 words
        for (size_t k = 0; k < batch; k++)
          HIPCHK(hipMemcpyPeerAsync(slice + k * p.n * 8, D->device, d_scalars + (k * stride + p.lo) * 8, P.device, p.n * 32, D->fan));
        uint32_t* part = (uint32_t*)((char*)dsc->small.p + 4096);
        if ((r = prepared_batch_enqueue(dsc, D->fan, p.pb, p.pb_off, slice, p.n, batch, p.n, part)) != ZKHIP_OK) return r;
        HIPCHK(hipMemcpyPeerAsync(gather + i * batch * 24, P.device, part, D->device, batch * 96, D->fan));
      }
      return ZKHIP_OK;
    };
    rc_remote = body();
    if (D->fan_done && D->fan) { (void)hipEventRecord(D->fan_done, D->fan); waited.push_back(d); }   // also after an error: s must not run ahead of work already queued
  }
  if (hipSetDevice(P.device) != hipSuccess) { set_error("hipSetDevice(%d) failed", P.device); return ZKHIP_ENODEV; }
  int rc_local = ZKHIP_OK;
  {
    size_t ws_bytes = 0;
    for (auto& p : pieces) if (p.dev == 0) ws_bytes = std::max(ws_bytes, msm_workspace_bytes(p.n, p.pb->c, true, batch_group(p.pb, p.n, batch)));
    rc_local = sc->ws.reserve(ws_bytes);
  }
  for (size_t i = 0; i < np && rc_local == ZKHIP_OK && rc_remote == ZKHIP_OK; i++) {
    const piece_t& p = pieces[i];
    if (p.dev != 0) continue;
    rc_local = prepared_batch_enqueue(sc, s, p.pb, p.pb_off, d_scalars + p.lo * 8, p.n, batch, stride, gather + i * batch * 24);
  }
  for (size_t d : waited) HIPCHK(hipStreamWaitEvent(s, g_ctx.devs[d]->fan_done, 0));
  if (rc_remote != ZKHIP_OK) return rc_remote;
  if (rc_local != ZKHIP_OK) return rc_local;
  return sum_jacobian_device(gather, (int)np, d_out, s, batch);
}

}  // namespace zkhip

extern "C" {

// device-resident scalars against bases pinned with zkhip_register_bases (`bases` = a host pointer into a registered array): the commit
// of a polynomial that lives in HBM against the SRS the host registered, without a second table (zkhip_prepare_bases_device).  A range
// that spans several shards fans out over their devices (registered_device_msm above).
int zkhip_msm_g1_registered_batch_device(const uint64_t* bases, const void* d_scalars, size_t n, size_t batch, size_t scalar_stride, void* d_out_xyz,
                                         void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (batch == 0) return ZKHIP_OK;
  if (!d_out_xyz || (n && (!d_scalars || !bases)) || (batch > 1 && scalar_stride < n)) { set_error("msm_registered: bad argument"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  if (n == 0) return msm_g1_device(nullptr, nullptr, 0, (uint32_t*)d_out_xyz, nullptr, 0, 0, s, nullptr, 0, batch, 0);
  size_t off = 0;
  std::shared_ptr<registered_t> reg = find_registered(bases, n, &off);
  if (!reg) { set_error("msm_registered: the range is not inside a registered array"); return ZKHIP_EINVAL; }
  // the tables outlive the enqueued work: zkhip_unregister_bases synchronises every device before the last reference goes
  return registered_device_msm(reg, off, (const uint32_t*)d_scalars, n, batch, scalar_stride, (uint32_t*)d_out_xyz, s);
}

int zkhip_msm_g1_registered_device(const uint64_t* bases, const void* d_scalars, size_t n, void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  return zkhip_msm_g1_registered_batch_device(bases, d_scalars, n, 1, n, d_out_xyz, stream);
}

int zkhip_msm_g1_prepared_batch_device(uint64_t handle, size_t offset, const void* d_scalars, size_t n, size_t batch, size_t scalar_stride,
                                       void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("msm_prepared_batch: unknown handle"); return ZKHIP_EINVAL; }
  if (!d_out_xyz || (n && batch && !d_scalars) || (batch > 1 && scalar_stride < n)) { set_error("msm_prepared_batch: bad argument"); return ZKHIP_EINVAL; }
  const prepared_bases* pb = it->second;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  // tables built for wide windows (n >= 2^20) do not batch: those MSMs are throughput-bound one at a time
  size_t group = (pb->c > 16 || batch <= 1) ? 1 : batch;
  while (group > 1 && ((size_t)((256 + pb->c - 1) / pb->c) * n * group >= (1ull << 31) || (group << (pb->c - 1)) > (1ull << 22))) group = (group + 1) / 2;   // <= 4 Mi buckets per launch set
  if (group == 1 && batch > 1 && n >= ((size_t)1 << 18)) {
    // Large MSMs do not share a launch set, but two of them overlap: while one is in its latency-bound sort and reduction tail the
    // other's accumulation fills the chip (measured: 2^20 1.50 -> 1.30 ms, 2^22 5.26 -> 4.87 ms per MSM).  Odd vectors go to a
    // library-owned high-priority stream (streams of different priority never share a hardware queue) forked from / joined to the
    // caller's stream with events; each stream has its own scratch set.
    device_ctx& P = primary();
    if (!P.side) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // hi = numerically lowest = greatest priority
      HIPCHK(hipStreamCreateWithPriority(&P.side, hipStreamNonBlocking, hi));
      HIPCHK(hipEventCreateWithFlags(&P.side_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&P.side_join, hipEventDisableTiming));
    }
    scratch* sc2 = scratch_for(P, P.side);
    const size_t ws_bytes = msm_workspace_bytes(n, pb->c, true, 1);
    if ((rc = sc->ws.reserve(ws_bytes)) != ZKHIP_OK) return rc;
    if ((rc = sc2->ws.reserve(ws_bytes)) != ZKHIP_OK) return rc;
    HIPCHK(hipEventRecord(P.side_fork, s));
    HIPCHK(hipStreamWaitEvent(P.side, P.side_fork, 0));
    for (size_t k0 = 0; k0 < batch; k0++) {
      const bool on_side = (k0 & 1) != 0;
      scratch* use = on_side ? sc2 : sc;
      rc = msm_g1_device((const uint32_t*)d_scalars + k0 * scalar_stride * 8, nullptr, n, (uint32_t*)d_out_xyz + k0 * 24, use->ws.p, use->ws.cap, 0, on_side ? P.side : s, pb,
                         offset, 1, scalar_stride);
      if (rc != ZKHIP_OK) break;
    }
    HIPCHK(hipEventRecord(P.side_join, P.side));               // also on an error: the caller's stream must not run ahead of the side stream
    HIPCHK(hipStreamWaitEvent(s, P.side_join, 0));
    return rc;
  }
  for (size_t k0 = 0; k0 < batch; k0 += group) {
    const size_t kk = batch - k0 < group ? batch - k0 : group;
    if ((rc = sc->ws.reserve(msm_workspace_bytes(n, pb->c, true, kk))) != ZKHIP_OK) return rc;
    rc = msm_g1_device((const uint32_t*)d_scalars + k0 * scalar_stride * 8, nullptr, n, (uint32_t*)d_out_xyz + k0 * 24, sc->ws.p, sc->ws.cap, 0, s, pb,
                       offset, kk, scalar_stride);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

int zkhip_g1_sum_device(const void* d_points_xyz, int m, void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (m < 0 || !d_out_xyz || (m && !d_points_xyz)) { set_error("g1_sum: bad argument"); return ZKHIP_EINVAL; }
  return sum_jacobian_device((const uint32_t*)d_points_xyz, m, (uint32_t*)d_out_xyz, caller_stream(stream));
}

int zkhip_g1_sum(const uint64_t* points_xyz, int m, uint64_t out_xyz[12]) {
  ZK_API_RANGE();
  if (m < 0 || !out_xyz || (m && !points_xyz)) { set_error("g1_sum: bad argument"); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  if ((rc = H.sc->small.reserve(4096 + (size_t)m * 96)) != ZKHIP_OK) return rc;
  char* d = (char*)H.sc->small.p;
  if (m) HIPCHK(hipMemcpyAsync(d + 4096, points_xyz, (size_t)m * 96, hipMemcpyHostToDevice, H.s));
  if ((rc = sum_jacobian_device((const uint32_t*)(d + 4096), m, (uint32_t*)d, H.s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(H.L->pinned, d, 96, hipMemcpyDeviceToHost, H.s));
  HIPCHK(hipStreamSynchronize(H.s));
  memcpy(out_xyz, H.L->pinned, 96);
  return ZKHIP_OK;
}

}  // extern "C"

namespace zkhip {

// ---- NTT / domain ----------------------------------------------------------------------------------
// runs ntt_transform with scratch from the stream's set (two buffers of batch * 2^L elements when the transform is multi-pass)
static int run_transform(scratch* sc, const uint32_t* d_in, uint32_t in_len, uint32_t in_stride, uint32_t* d_out, uint32_t out_len, uint32_t out_stride,
                         uint32_t batch, uint32_t L, const uint32_t* omega, const uint32_t* in_scale, uint32_t in_period,
                         const uint32_t* out_scale, uint32_t out_period, hipStream_t s) {
  if (L > 28) { set_error("ntt: log_n = %u > 28", L); return ZKHIP_EINVAL; }
  const int np = ntt_passes(L);
  // the passes ping-pong through one or two scratch copies of the batch: cap the scratch at ~2 GiB per copy by transforming a large
  // batch in sub-batches (same kernels, same throughput; 26 polynomials of 2^24 would otherwise need 28 GB of scratch)
  uint32_t sub = batch;
  const size_t cap = (size_t)1 << 31;
  if (batch > 1 && (((size_t)batch << L) * 32) > cap) sub = (uint32_t)std::max<size_t>(1, cap / (((size_t)1 << L) * 32));
  const size_t one = (size_t)sub << L;     // elements per scratch copy
  uint32_t *t0 = nullptr, *t1 = nullptr;
  if (np >= 2) {
    int rc = sc->ntt_tmp.reserve(one * 32 * (np >= 3 ? 2 : 1));
    if (rc != ZKHIP_OK) return rc;
    t0 = (uint32_t*)sc->ntt_tmp.p;
    if (np >= 3) t1 = t0 + one * 8;
  }
  for (uint32_t b0 = 0; b0 < batch; b0 += sub) {
    const uint32_t nb = batch - b0 < sub ? batch - b0 : sub;
    int rc = ntt_transform(d_in + (size_t)b0 * in_stride * 8, in_len, in_stride, d_out + (size_t)b0 * out_stride * 8, out_len, out_stride, nb, L, omega,
                           in_scale, in_period, out_scale, out_period, t0, t1, s);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

// a `_device` transform on the caller's stream (g_mu held by the caller)
static int device_transform_one(const void* d_in, uint32_t in_len, size_t in_stride, void* d_out, uint32_t out_len, size_t out_stride, uint32_t batch, uint32_t L,
                                const uint64_t* omega, const uint32_t* in_scale, uint32_t in_period, const uint32_t* out_scale, uint32_t out_period, hipStream_t s) {
  return run_transform(scratch_for(primary(), s), (const uint32_t*)d_in, in_len, (uint32_t)in_stride, (uint32_t*)d_out, out_len, (uint32_t)out_stride, batch, L,
                       (const uint32_t*)omega, in_scale, in_period, out_scale, out_period, s);
}

// Round 5 (second split): the polynomials of a batch are independent units, so a batched transform may be spread over the
// devices of zkhip_init -- every transform still runs on ONE device.  Polynomials [lo_d, hi_d) (shard_range over the batch) go to device d.
// `_device` form (this function, fan-out mode 2 only): device d > 0 waits for the caller's stream (fan_ready), pulls its polynomials from the
// primary's HBM over xGMI (hipMemcpyPeerAsync) into its own scratch, transforms them on its `fan` stream with its own twiddle plan, pushes the
// results back into d_out and records fan_done, which the caller's stream waits for.  Asynchronous like every `_device` call.
// Off by default: a 2^24 result is 512 MiB over one xGMI link (about 8 ms at 65 GB/s) against a 1.8 ms transform, so copying out and back
// only pays where the links are idle and the batch is deep; the host-buffer forms below (mode 1, default) are the ones that win outright --
// each device moves its own polynomials over its own PCIe link.  Results are identical either way (one device per transform).
static int device_transform(const void* d_in, uint32_t in_len, size_t in_stride, void* d_out, uint32_t out_len, size_t out_stride, uint32_t batch, uint32_t L,
                            const uint64_t* omega, const uint32_t* in_scale, uint32_t in_period, const uint32_t* out_scale, uint32_t out_period, void* stream) {
  hipStream_t s = caller_stream(stream);
  const int S = (int)g_ctx.devs.size();
  if (g_ctx.ntt_fanout < 2 || S < 2 || batch < 2 || L < 12)
    return device_transform_one(d_in, in_len, in_stride, d_out, out_len, out_stride, batch, L, omega, in_scale, in_period, out_scale, out_period, s);
  device_ctx& P = primary();
  const size_t N = (size_t)1 << L;
  if (!P.fan_ready) HIPCHK(hipEventCreateWithFlags(&P.fan_ready, hipEventDisableTiming));
  HIPCHK(hipEventRecord(P.fan_ready, s));
  int rc_remote = ZKHIP_OK;
  std::vector<int> waited;
  for (int d = 1; d < S && rc_remote == ZKHIP_OK; d++) {
    size_t lo, hi;
    shard_range(batch, d, S, &lo, &hi);
    if (lo >= hi) continue;
    device_ctx* D = g_ctx.devs[(size_t)d];
    if (hipSetDevice(D->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", D->device); rc_remote = ZKHIP_ENODEV; break; }
    auto body = [&]() -> int {
      if (!D->fan) {
        HIPCHK(hipStreamCreateWithFlags(&D->fan, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&D->fan_done, hipEventDisableTiming));
      }
      scratch* dsc = scratch_for(*D, D->fan);
      const size_t cnt = hi - lo;
      int r;
      if ((r = dsc->poly.reserve(cnt * N * 32)) != ZKHIP_OK) return r;
      HIPCHK(hipStreamWaitEvent(D->fan, P.fan_ready, 0));
      uint32_t* buf = (uint32_t*)dsc->poly.p;                 // polynomial b of this device at buf + b * N elements, transformed in place
      for (size_t b = 0; b < cnt; b++)
        HIPCHK(hipMemcpyPeerAsync(buf + b * N * 8, D->device, (const uint32_t*)d_in + (lo + b) * in_stride * 8, P.device, (size_t)in_len * 32, D->fan));
      if ((r = run_transform(dsc, buf, in_len, (uint32_t)N, buf, out_len, (uint32_t)N, (uint32_t)cnt, L, (const uint32_t*)omega, in_scale, in_period,
                             out_scale, out_period, D->fan)) != ZKHIP_OK) return r;
      for (size_t b = 0; b < cnt; b++)
        HIPCHK(hipMemcpyPeerAsync((uint32_t*)d_out + (lo + b) * out_stride * 8, P.device, buf + b * N * 8, D->device, (size_t)out_len * 32, D->fan));
      return ZKHIP_OK;
    };
    rc_remote = body();
    if (D->fan_done && D->fan) { (void)hipEventRecord(D->fan_done, D->fan); waited.push_back(d); }   // also after an error: s must not run ahead of work already queued
  }
  if (hipSetDevice(P.device) != hipSuccess) { set_error("hipSetDevice(%d) failed", P.device); return ZKHIP_ENODEV; }
  int rc_local = ZKHIP_OK;
  {
    size_t lo, hi;
    shard_range(batch, 0, S, &lo, &hi);
    if (lo < hi && rc_remote == ZKHIP_OK)
      rc_local = device_transform_one((const uint32_t*)d_in + lo * in_stride * 8, in_len, in_stride, (uint32_t*)d_out + lo * out_stride * 8, out_len, out_stride,
                                      (uint32_t)(hi - lo), L, omega, in_scale, in_period, out_scale, out_period, s);
  }
  for (int d : waited) HIPCHK(hipStreamWaitEvent(s, g_ctx.devs[(size_t)d]->fan_done, 0));
  return rc_remote != ZKHIP_OK ? rc_remote : rc_local;
}

}  // namespace zkhip

extern "C" {

int zkhip_ntt_fr_batch_device(void* d_a, const uint64_t omega[4], uint32_t log_n, uint32_t batch, size_t stride, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega || log_n > 28 || stride < ((size_t)1 << log_n) || stride >= ((size_t)1 << 32)) { set_error("ntt_batch: bad argument"); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return device_transform(d_a, N, stride, d_a, N, stride, batch, log_n, omega, nullptr, 0, nullptr, 0, stream);
}

int zkhip_ifft_scaled_batch_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], uint32_t batch, size_t stride,
                                   void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega_inv || !divisor || log_n > 28 || stride < ((size_t)1 << log_n) || stride >= ((size_t)1 << 32)) { set_error("ifft_batch: bad argument"); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return device_transform(d_a, N, stride, d_a, N, stride, batch, log_n, omega_inv, nullptr, 0, (const uint32_t*)divisor, 1, stream);
}

int zkhip_ntt_fr_device(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega) { set_error("ntt: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 28) { set_error("ntt: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return device_transform(d_a, N, N, d_a, N, N, 1, log_n, omega, nullptr, 0, nullptr, 0, stream);
}

int zkhip_ifft_scaled_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega_inv || !divisor) { set_error("ifft: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 28) { set_error("ifft: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return device_transform(d_a, N, N, d_a, N, N, 1, log_n, omega_inv, nullptr, 0, (const uint32_t*)divisor, 1, stream);
}

int zkhip_mul_periodic_device(void* d_a, size_t n, const void* d_table, uint32_t period, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((n && !d_a) || !d_table) { set_error("mul_periodic: null pointer"); return ZKHIP_EINVAL; }
  return fr_mul_periodic_device((uint32_t*)d_a, n, (const uint32_t*)d_table, period, caller_stream(stream));
}

}  // extern "C"

namespace zkhip {

// One device's share of a host-buffer batch: `cnt` polynomials (polynomial b at in + b * in_stride elements) cross this device's PCIe link
// into its scratch, are transformed in place (polynomial b at b * N elements) and go back to out + b * out_stride.  Runs on the calling
// thread for the primary device and on the device's worker thread for a secondary (the current device is per host thread).
static int host_transform_share(scratch* sc, hipStream_t s, const uint64_t* in, size_t in_len, size_t in_stride, uint64_t* out, size_t out_len, size_t out_stride,
                                size_t cnt, uint32_t log_n, const uint64_t* omega, const uint32_t* in_scale, uint32_t in_period, const uint32_t* out_scale,
                                uint32_t out_period) {
  int rc;
  const size_t N = (size_t)1 << log_n;
  if ((rc = sc->poly.reserve(cnt * N * 32)) != ZKHIP_OK) return rc;
  if (cnt == 1) HIPCHK(hipMemcpyAsync(sc->poly.p, in, in_len * 32, hipMemcpyHostToDevice, s));
  else if (in_len == N && in_stride == N) HIPCHK(hipMemcpyAsync(sc->poly.p, in, cnt * N * 32, hipMemcpyHostToDevice, s));
  else HIPCHK(hipMemcpy2DAsync(sc->poly.p, N * 32, in, in_stride * 32, in_len * 32, cnt, hipMemcpyHostToDevice, s));
  rc = run_transform(sc, (const uint32_t*)sc->poly.p, (uint32_t)in_len, (uint32_t)N, (uint32_t*)sc->poly.p, (uint32_t)out_len, (uint32_t)N, (uint32_t)cnt, log_n,
                     (const uint32_t*)omega, in_scale, in_period, out_scale, out_period, s);
  if (rc != ZKHIP_OK) return rc;
  if (cnt == 1) HIPCHK(hipMemcpyAsync(out, sc->poly.p, out_len * 32, hipMemcpyDeviceToHost, s));
  else if (out_len == N && out_stride == N) HIPCHK(hipMemcpyAsync(out, sc->poly.p, cnt * N * 32, hipMemcpyDeviceToHost, s));
  else HIPCHK(hipMemcpy2DAsync(out, out_stride * 32, sc->poly.p, N * 32, out_len * 32, cnt, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// Host-buffer transforms, `batch` polynomials.  With several devices (fan-out mode >= 1) the batch is cut by shard_range: the primary's share
// runs on this thread's lane, every other device's share on its worker thread -- N PCIe links carry the batch instead of one (a 2^24
// transform through ONE link is 21 ms around a 1.8 ms kernel).  Each transform runs on one device: the results do not depend on the split.
static int host_transform(const uint64_t* in, size_t in_len, size_t in_stride, uint64_t* out, size_t out_len, size_t out_stride, uint32_t batch, uint32_t log_n,
                          const uint64_t* omega, const uint32_t* in_scale, uint32_t in_period, const uint32_t* out_scale, uint32_t out_period) {
  if (log_n > 28) { set_error("ntt: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  if (!in || !out || !omega) { set_error("ntt: null pointer"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << log_n;
  if (in_len > N || out_len > N || (batch > 1 && (in_stride < in_len || out_stride < out_len))) { set_error("ntt: length exceeds domain or stride"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int S, mode;
  { guard_t g(g_mu); S = (int)g_ctx.devs.size(); mode = g_ctx.ntt_fanout; }
  if (S < 2 || batch < 2 || mode < 1)
    return host_transform_share(H.sc, H.s, in, in_len, in_stride, out, out_len, out_stride, batch, log_n, omega, in_scale, in_period, out_scale, out_period);
  std::unique_lock<std::mutex> fan(g_fanout_mu);              // the secondary devices have one lane each
  for (int d = 1; d < S; d++) {
    size_t lo, hi;
    shard_range(batch, d, S, &lo, &hi);
    if (lo >= hi) continue;
    device_ctx* D = g_ctx.devs[(size_t)d];
    D->w->submit([=]() -> int {
      hipStream_t ds = D->lanes[0].stream;
      scratch* dsc;
      { guard_t g(g_mu); dsc = scratch_for(*D, ds); }
      return host_transform_share(dsc, ds, in + lo * in_stride * 4, in_len, in_stride, out + lo * out_stride * 4, out_len, out_stride, hi - lo, log_n, omega,
                                  in_scale, in_period, out_scale, out_period);
    });
  }
  int rc_local = ZKHIP_OK;
  {
    size_t lo, hi;
    shard_range(batch, 0, S, &lo, &hi);
    if (lo < hi)
      rc_local = host_transform_share(H.sc, H.s, in + lo * in_stride * 4, in_len, in_stride, out + lo * out_stride * 4, out_len, out_stride, hi - lo, log_n, omega,
                                      in_scale, in_period, out_scale, out_period);
  }
  int rc_remote = ZKHIP_OK;
  for (int d = 1; d < S; d++) {
    size_t lo, hi;
    shard_range(batch, d, S, &lo, &hi);
    if (lo >= hi) continue;
    const int r = g_ctx.devs[(size_t)d]->w->wait();
    if (r != ZKHIP_OK) rc_remote = r;
  }
  return rc_local != ZKHIP_OK ? rc_local : rc_remote;
}

}  // namespace zkhip

extern "C" {

int zkhip_set_ntt_fanout(int mode) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (mode < 0 || mode > 2) { set_error("set_ntt_fanout: mode %d not in 0..2", mode); return ZKHIP_EINVAL; }
  g_ctx.ntt_fanout = mode;
  return ZKHIP_OK;
}

int zkhip_ntt_fanout(void) {
  guard_t g(g_mu);
  return g_ctx.ready ? g_ctx.ntt_fanout : 1;
}

// `batch` contiguous polynomials of 2^log_n elements, transformed in place (one launch set per device)
int zkhip_ntt_fr_batch(uint64_t* a, const uint64_t omega[4], uint32_t log_n, uint32_t batch) {
  ZK_API_RANGE();
  if (!a || !omega || log_n > 28) { set_error("ntt_batch: bad argument"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << log_n;
  return host_transform(a, N, N, a, N, N, batch, log_n, omega, nullptr, 0, nullptr, 0);
}

int zkhip_ifft_scaled_batch(uint64_t* a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], uint32_t batch) {
  ZK_API_RANGE();
  if (!a || !omega_inv || !divisor || log_n > 28) { set_error("ifft_batch: bad argument"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << log_n;
  return host_transform(a, N, N, a, N, N, batch, log_n, omega_inv, nullptr, 0, (const uint32_t*)divisor, 1);
}

int zkhip_ntt_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
  ZK_API_RANGE();
  const size_t N = (size_t)1 << (log_n > 28 ? 0 : log_n);
  return host_transform(a, N, N, a, N, N, 1, log_n, omega, nullptr, 0, nullptr, 0);
}

int zkhip_ifft_scaled(uint64_t* a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]) {
  ZK_API_RANGE();
  if (!divisor) { set_error("ifft: null divisor"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << (log_n > 28 ? 0 : log_n);
  return host_transform(a, N, N, a, N, N, 1, log_n, omega_inv, nullptr, 0, (const uint32_t*)divisor, 1);
}

}  // extern "C"

namespace zkhip {
// scale triples of the coset transforms, computed on the host (4 x 64 Montgomery, include/zkhip.hpp):
//   into the coset:   {1, zeta, zeta^2}                      (distribute_powers_zeta(.., true): [g_coset, g_coset_inv])
//   out of the coset: divisor * {1, zeta^2, zeta}            (distribute_powers_zeta(.., false) + the ifft divisor)
static void coset_scales(const uint64_t zeta[4], const uint64_t* divisor, uint32_t out[24]) {
  namespace hd = zkhip::halo2::detail;
  zkhip::halo2::Fr z, zz, c0;
  memcpy(z.l, zeta, 32);
  zz = hd::mul(z, z);
  if (!divisor) {
    c0 = hd::one();
    memcpy(out, c0.l, 32); memcpy(out + 8, z.l, 32); memcpy(out + 16, zz.l, 32);
  } else {
    memcpy(c0.l, divisor, 32);
    const zkhip::halo2::Fr c1 = hd::mul(c0, zz), c2 = hd::mul(c0, z);
    memcpy(out, c0.l, 32); memcpy(out + 8, c1.l, 32); memcpy(out + 16, c2.l, 32);
  }
}
}  // namespace zkhip

extern "C" {

int zkhip_coeff_to_extended(const uint64_t* a, uint32_t k, uint64_t* out, uint32_t ext_k, const uint64_t ext_omega[4], const uint64_t zeta[4]) {
  ZK_API_RANGE();
  if (!a || !out || !ext_omega || !zeta || k > ext_k || ext_k > 28) { set_error("coeff_to_extended: bad argument"); return ZKHIP_EINVAL; }
  uint32_t sc[24];
  coset_scales(zeta, nullptr, sc);
  return host_transform(a, (size_t)1 << k, (size_t)1 << k, out, (size_t)1 << ext_k, (size_t)1 << ext_k, 1, ext_k, ext_omega, sc, 3, nullptr, 0);
}

// `batch` polynomials: polynomial b = a[b * 2^k ..], its extended-coset evaluations = out[b * 2^ext_k ..] (the per-column loop of the
// quotient phase in one call; with several devices each takes its share of the columns over its own PCIe link)
int zkhip_coeff_to_extended_batch(const uint64_t* a, uint32_t k, uint64_t* out, uint32_t ext_k, uint32_t batch, const uint64_t ext_omega[4],
                                  const uint64_t zeta[4]) {
  ZK_API_RANGE();
  if (!a || !out || !ext_omega || !zeta || k > ext_k || ext_k > 28) { set_error("coeff_to_extended_batch: bad argument"); return ZKHIP_EINVAL; }
  uint32_t sc[24];
  coset_scales(zeta, nullptr, sc);
  return host_transform(a, (size_t)1 << k, (size_t)1 << k, out, (size_t)1 << ext_k, (size_t)1 << ext_k, batch, ext_k, ext_omega, sc, 3, nullptr, 0);
}

int zkhip_extended_to_coeff(uint64_t* a, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                            const uint64_t zeta[4], uint64_t* out, size_t out_len) {
  ZK_API_RANGE();
  if (!a || !out || !ext_omega_inv || !ext_divisor || !zeta || ext_k > 28) { set_error("extended_to_coeff: bad argument"); return ZKHIP_EINVAL; }
  uint32_t sc[24];
  coset_scales(zeta, ext_divisor, sc);
  return host_transform(a, (size_t)1 << ext_k, (size_t)1 << ext_k, out, out_len, out_len, 1, ext_k, ext_omega_inv, nullptr, 0, sc, 3);
}

// device-resident forms, `batch` polynomials per launch set (polynomial b at base + b * stride elements)
int zkhip_coeff_to_extended_device(const void* d_a, size_t a_stride, uint32_t k, void* d_out, size_t out_stride, uint32_t ext_k, uint32_t batch,
                                   const uint64_t ext_omega[4], const uint64_t zeta[4], void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !d_out || !ext_omega || !zeta || k > ext_k || ext_k > 28 || a_stride < ((size_t)1 << k) || out_stride < ((size_t)1 << ext_k) ||
      a_stride >= ((size_t)1 << 32) || out_stride >= ((size_t)1 << 32)) { set_error("coeff_to_extended: bad argument"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  uint32_t sc[24];
  coset_scales(zeta, nullptr, sc);
  return device_transform(d_a, 1u << k, a_stride, d_out, 1u << ext_k, out_stride, batch, ext_k, ext_omega, sc, 3, nullptr, 0, stream);
}

int zkhip_extended_to_coeff_device(const void* d_a, size_t a_stride, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                                   const uint64_t zeta[4], void* d_out, size_t out_stride, size_t out_len, uint32_t batch, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !d_out || !ext_omega_inv || !ext_divisor || !zeta || ext_k > 28 || out_len > ((size_t)1 << ext_k) || a_stride < ((size_t)1 << ext_k) ||
      out_stride < out_len || a_stride >= ((size_t)1 << 32) || out_stride >= ((size_t)1 << 32)) { set_error("extended_to_coeff: bad argument"); return ZKHIP_EINVAL; }
  if (batch == 0 || out_len == 0) return ZKHIP_OK;
  uint32_t sc[24];
  coset_scales(zeta, ext_divisor, sc);
  return device_transform(d_a, 1u << ext_k, a_stride, d_out, (uint32_t)out_len, out_stride, batch, ext_k, ext_omega_inv, nullptr, 0, sc, 3, stream);
}

int zkhip_mul_periodic(uint64_t* a, size_t n, const uint64_t* table, uint32_t period) {
  ZK_API_RANGE();
  if ((n && !a) || !table || period == 0) { set_error("mul_periodic: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * 32)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve((size_t)period * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(H.sc->poly.p, a, n * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(H.sc->poly2.p, table, (size_t)period * 32, hipMemcpyHostToDevice, s));
  if ((rc = fr_mul_periodic_device((uint32_t*)H.sc->poly.p, n, (const uint32_t*)H.sc->poly2.p, period, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(a, H.sc->poly.p, n * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// ---- row a7: Fr-vector primitives ---------------------------------------------------------------------
int zkhip_fr_eval_polynomial_device(const void* d_poly, size_t n, const uint64_t point[4], void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!point || !d_out || (n && !d_poly)) { set_error("eval_polynomial: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_eval_polynomial_device((const uint32_t*)d_poly, n, (const uint32_t*)point, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s);
}

int zkhip_fr_eval_polynomial_batch_device(const void* const* d_polys, size_t count, size_t n, const uint64_t point[4], void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (count && (!d_polys || !point || !d_out)) { set_error("eval_polynomial_batch: null pointer"); return ZKHIP_EINVAL; }
  for (size_t i = 0; i < count && n; i++)
    if (!d_polys[i]) { set_error("eval_polynomial_batch: polynomial %zu is null", i); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(poly_batch_workspace_bytes(n, count))) != ZKHIP_OK) return rc;
  return fr_eval_polynomial_batch_device(d_polys, count, n, (const uint32_t*)point, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s, &sc->args);
}

int zkhip_fr_kate_division_device(const void* d_a, size_t n, const uint64_t b[4], void* d_q, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!b || (n > 1 && (!d_a || !d_q))) { set_error("kate_division: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_kate_division_device((const uint32_t*)d_a, n, (const uint32_t*)b, (uint32_t*)d_q, sc->ws.p, sc->ws.cap, s);
}

int zkhip_fr_batch_invert_device(void* d_a, size_t n, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && !d_a) { set_error("batch_invert: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_batch_invert_device((uint32_t*)d_a, n, sc->ws.p, sc->ws.cap, s);
}

int zkhip_fr_prefix_product_device(const void* d_v, size_t n, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_v || !d_out)) { set_error("prefix_product: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_prefix_product_device((const uint32_t*)d_v, n, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s);
}

}  // extern "C"

namespace zkhip {
// host-buffer wrappers: upload to the lane's poly scratch, run on the lane's stream, download
static int host_vec_op(int op, const uint64_t* in, size_t n_in, const uint64_t* c, uint64_t* out, size_t n_out) {
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve((n_in + 1) * 32)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve((n_out + 1) * 32)) != ZKHIP_OK) return rc;
  if (n_in) HIPCHK(hipMemcpyAsync(H.sc->poly.p, in, n_in * 32, hipMemcpyHostToDevice, s));
  switch (op) {
    case 0: rc = zkhip_fr_eval_polynomial_device(H.sc->poly.p, n_in, c, H.sc->poly2.p, s); break;
    case 1: rc = zkhip_fr_kate_division_device(H.sc->poly.p, n_in, c, H.sc->poly2.p, s); break;
    case 2: rc = zkhip_fr_batch_invert_device(H.sc->poly.p, n_in, s); break;
    default: rc = zkhip_fr_prefix_product_device(H.sc->poly.p, n_in, H.sc->poly2.p, s); break;
  }
  if (rc != ZKHIP_OK) return rc;
  if (n_out) HIPCHK(hipMemcpyAsync(out, op == 2 ? H.sc->poly.p : H.sc->poly2.p, n_out * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}
}  // namespace zkhip

extern "C" {

int zkhip_fr_eval_polynomial(const uint64_t* poly, size_t n, const uint64_t point[4], uint64_t out[4]) {
  ZK_API_RANGE();
  if (!point || !out || (n && !poly)) { set_error("eval_polynomial: null pointer"); return ZKHIP_EINVAL; }
  return host_vec_op(0, poly, n, point, out, 1);
}

int zkhip_fr_kate_division(const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* q) {
  ZK_API_RANGE();
  if (!b || (n > 1 && (!a || !q))) { set_error("kate_division: null pointer"); return ZKHIP_EINVAL; }
  if (n < 2) return ZKHIP_OK;
  return host_vec_op(1, a, n, b, q, n - 1);
}

int zkhip_fr_batch_invert(uint64_t* a, size_t n) {
  ZK_API_RANGE();
  if (n && !a) { set_error("batch_invert: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  return host_vec_op(2, a, n, nullptr, a, n);
}

int zkhip_fr_prefix_product(const uint64_t* v, size_t n, uint64_t* out) {
  ZK_API_RANGE();
  if (n && (!v || !out)) { set_error("prefix_product: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  return host_vec_op(3, v, n, nullptr, out, n);
}

// ---- lookup argument: permute_expression_pair ----------------------------------------------------------------------
int zkhip_lookup_permute_device(const void* d_input, const void* d_table, size_t usable_rows, void* d_permuted_input, void* d_permuted_table,
                                void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (usable_rows && (!d_input || !d_table || !d_permuted_input || !d_permuted_table)) { set_error("lookup_permute: null pointer"); return ZKHIP_EINVAL; }
  if (usable_rows == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->vm.reserve(lookup_permute_workspace_bytes(usable_rows))) != ZKHIP_OK) return rc;
  return lookup_permute_device((const uint32_t*)d_input, (const uint32_t*)d_table, usable_rows, (uint32_t*)d_permuted_input,
                               (uint32_t*)d_permuted_table, sc->vm.p, sc->vm.cap, s);
}

int zkhip_lookup_permute(const uint64_t* input, const uint64_t* table, size_t usable_rows, uint64_t* permuted_input, uint64_t* permuted_table) {
  ZK_API_RANGE();
  if (usable_rows && (!input || !table || !permuted_input || !permuted_table)) { set_error("lookup_permute: null pointer"); return ZKHIP_EINVAL; }
  if (usable_rows == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  const size_t bytes = usable_rows * 32;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(2 * bytes)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve(2 * bytes)) != ZKHIP_OK) return rc;
  char* in = (char*)H.sc->poly.p;
  char* out = (char*)H.sc->poly2.p;
  HIPCHK(hipMemcpyAsync(in, input, bytes, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(in + bytes, table, bytes, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_lookup_permute_device(in, in + bytes, usable_rows, out, out + bytes, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(permuted_input, out, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(permuted_table, out + bytes, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// ---- device buffers for hosts that do not link HIP -------------------------------------------------------------
int zkhip_alloc(size_t bytes, void** d_ptr) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_ptr) { set_error("alloc: null pointer"); return ZKHIP_EINVAL; }
  *d_ptr = nullptr;
  if (bytes == 0) return ZKHIP_OK;
  if (hipMalloc(d_ptr, bytes) != hipSuccess) { (void)hipGetLastError(); *d_ptr = nullptr; set_error("hipMalloc(%zu) failed", bytes); return ZKHIP_ENOMEM; }
  return ZKHIP_OK;
}

int zkhip_free(void* d_ptr) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  if (!d_ptr) return ZKHIP_OK;
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipDeviceSynchronize());               // kernels queued on any stream may still use it
  HIPCHK(hipFree(d_ptr));
  return ZKHIP_OK;
}

int zkhip_upload(void* d_dst, const void* src, size_t bytes) {
  ZK_API_RANGE();
  { guard_t g(g_mu); int rc = ensure_init(); if (rc != ZKHIP_OK) return rc; }
  if (bytes && (!d_dst || !src)) { set_error("upload: null pointer"); return ZKHIP_EINVAL; }
  if (bytes == 0) return ZKHIP_OK;
  HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));      // default stream, blocking
  return ZKHIP_OK;
}

int zkhip_download(void* dst, const void* d_src, size_t bytes) {
  ZK_API_RANGE();
  { guard_t g(g_mu); int rc = ensure_init(); if (rc != ZKHIP_OK) return rc; }
  if (bytes && (!dst || !d_src)) { set_error("download: null pointer"); return ZKHIP_EINVAL; }
  if (bytes == 0) return ZKHIP_OK;
  HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return ZKHIP_OK;
}

int zkhip_stream_sync(void* stream) {
  ZK_API_RANGE();
  { guard_t g(g_mu); int rc = ensure_init(); if (rc != ZKHIP_OK) return rc; }
  HIPCHK(hipStreamSynchronize(caller_stream(stream)));
  return ZKHIP_OK;
}

int zkhip_sync(void) {
  ZK_API_RANGE();
  { guard_t g(g_mu); int rc = ensure_init(); if (rc != ZKHIP_OK) return rc; }
  HIPCHK(hipDeviceSynchronize());
  return ZKHIP_OK;
}

// ---- section 8(f): row programs and grand products -------------------------------------------------------------
int zkhip_fr_eval_rows_device(const zkhip_vm_program* prog, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows,
                              int accumulate, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((rc = row_vm_validate(prog, n_columns, log_rows, accumulate)) != ZKHIP_OK) return rc;
  if (!d_out || (n_columns && !d_columns)) { set_error("eval_rows: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->vm.reserve(row_vm_workspace_bytes(prog, n_columns, log_rows))) != ZKHIP_OK) return rc;
  return row_vm_device(prog, d_columns, n_columns, log_rows, accumulate, (uint32_t*)d_out, sc->vm.p, sc->vm.cap, s, &sc->vm_stage);
}

// out[row] = sum_p weights[p] * progs[p](row): independent row programs over the same columns in ONE launch (blockIdx.y = program:
// row_vm_device_multi), their outputs combined by the linear-combination kernel.  What it is for: the quotient numerator of a circuit with
// hundreds of columns at 2^13 .. 2^15 rows is thousands of instructions over a few thousand rows -- one program is a handful of wavefronts
// walking the whole list.  (Separate launches on side streams overlapped four at a time -- the hardware queues: 2.3 x where one grid gives 8 x.)
int zkhip_fr_eval_rows_sum_device(const zkhip_vm_program* progs, const uint64_t* weights, uint32_t n_progs, const void* const* d_columns, uint32_t n_columns,
                                  uint32_t log_rows, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!progs || !weights || n_progs == 0 || n_progs > 4096 || !d_out || (n_columns && !d_columns)) { set_error("eval_rows_sum: bad argument"); return ZKHIP_EINVAL; }
  for (uint32_t p = 0; p < n_progs; p++)
    if ((rc = row_vm_validate(&progs[p], n_columns, log_rows, 0)) != ZKHIP_OK) return rc;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  const size_t rows = (size_t)1 << log_rows;
  if ((rc = sc->poly2.reserve((size_t)n_progs * rows * 32)) != ZKHIP_OK) return rc;
  if ((rc = sc->vm.reserve(row_vm_multi_workspace_bytes(progs, n_progs, n_columns, log_rows))) != ZKHIP_OK) return rc;
  std::vector<uint32_t*> partial(n_progs);
  for (uint32_t p = 0; p < n_progs; p++) partial[p] = (uint32_t*)((char*)sc->poly2.p + (size_t)p * rows * 32);
  if ((rc = row_vm_device_multi(progs, n_progs, d_columns, n_columns, log_rows, partial.data(), sc->vm.p, sc->vm.cap, s, &sc->vm_stage)) != ZKHIP_OK) return rc;
  if ((rc = sc->ws.reserve(lincomb_workspace_bytes(n_progs, rows))) != ZKHIP_OK) return rc;
  return fr_linear_combination_device((const void* const*)partial.data(), (const uint32_t*)weights, n_progs, rows, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s, &sc->args);
}

// The straight-line HIP source rowvm_jit.hip generates for `prog` (buf may be NULL: *len receives the size needed, NUL included), and a
// compile-only run of it through hiprtc -- no device needed (the generator's CPU-side test; a host may also use it to warm hiprtc's caches at keygen).
int zkhip_vm_jit_source(const zkhip_vm_program* prog, uint32_t n_columns, uint32_t log_rows, char* buf, size_t cap, size_t* len) {
  ZK_API_RANGE();
  int rc = row_vm_validate(prog, n_columns, log_rows, 0);
  if (rc != ZKHIP_OK) return rc;
  std::string src;
  if ((rc = row_vm_jit_source(prog, n_columns, log_rows, &src)) != ZKHIP_OK) return rc;
  if (len) *len = src.size() + 1;
  if (buf && cap) { const size_t m = std::min(cap - 1, src.size()); memcpy(buf, src.data(), m); buf[m] = 0; }
  return ZKHIP_OK;
}

int zkhip_vm_jit_compile(const zkhip_vm_program* prog, uint32_t n_columns, uint32_t log_rows, size_t* code_bytes) {
  ZK_API_RANGE();
  int rc = row_vm_validate(prog, n_columns, log_rows, 0);
  if (rc != ZKHIP_OK) return rc;
  return row_vm_jit_compile_only(prog, n_columns, log_rows, code_bytes);
}

int zkhip_fr_eval_rows(const zkhip_vm_program* prog, const uint64_t* const* columns, uint32_t n_columns, uint32_t log_rows,
                       int accumulate, uint64_t* out) {
  ZK_API_RANGE();
  int rc;
  if ((rc = row_vm_validate(prog, n_columns, log_rows, accumulate)) != ZKHIP_OK) return rc;
  if (!out || (n_columns && !columns)) { set_error("eval_rows: null pointer"); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  const size_t rows = (size_t)1 << log_rows, bytes = rows * 32;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve((size_t)(n_columns ? n_columns : 1) * bytes)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve(bytes)) != ZKHIP_OK) return rc;
  std::vector<const void*> d_cols(n_columns);
  for (uint32_t i = 0; i < n_columns; i++) {
    if (!columns[i]) { set_error("eval_rows: column %u is null", i); return ZKHIP_EINVAL; }
    d_cols[i] = (char*)H.sc->poly.p + (size_t)i * bytes;
    HIPCHK(hipMemcpyAsync((void*)d_cols[i], columns[i], bytes, hipMemcpyHostToDevice, s));
  }
  if (accumulate) HIPCHK(hipMemcpyAsync(H.sc->poly2.p, out, bytes, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_fr_eval_rows_device(prog, d_cols.data(), n_columns, log_rows, accumulate, H.sc->poly2.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out, H.sc->poly2.p, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_fr_gather_mul_device(const void* d_a, size_t a_len, const void* d_index_a, const void* d_b, size_t b_len, const void* d_index_b, size_t n,
                               void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_a || !d_b || !d_index_a || !d_index_b || !d_out || a_len == 0 || b_len == 0 || a_len >= ((size_t)1 << 32) || b_len >= ((size_t)1 << 32))) {
    set_error("gather_mul: bad argument");
    return ZKHIP_EINVAL;
  }
  return fr_gather_mul_device((const uint32_t*)d_a, (uint32_t)a_len, (const uint32_t*)d_index_a, (const uint32_t*)d_b, (uint32_t)b_len, (const uint32_t*)d_index_b, n,
                              (uint32_t*)d_out, caller_stream(stream));
}

int zkhip_fr_grand_product_device(const void* d_num, void* d_den, size_t n, void* d_z, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_num || !d_den || !d_z)) { set_error("grand_product: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  if ((rc = zkhip_fr_batch_invert_device(d_den, n, s)) != ZKHIP_OK) return rc;
  if ((rc = fr_pointwise_mul_device((const uint32_t*)d_num, (const uint32_t*)d_den, n, (uint32_t*)d_den, s)) != ZKHIP_OK) return rc;
  return zkhip_fr_prefix_product_device(d_den, n, d_z, s);
}

int zkhip_fr_grand_product(const uint64_t* num, const uint64_t* den, size_t n, uint64_t* z) {
  ZK_API_RANGE();
  if (n && (!num || !den || !z)) { set_error("grand_product: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * 32)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve(n * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(H.sc->poly.p, num, n * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(H.sc->poly2.p, den, n * 32, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_fr_grand_product_device(H.sc->poly.p, H.sc->poly2.p, n, H.sc->poly.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(z, H.sc->poly.p, n * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_fr_linear_combination_device(const void* const* d_cols, const uint64_t* coeffs, size_t count, size_t n, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_out || (count && (!d_cols || !coeffs)))) { set_error("linear_combination: null pointer"); return ZKHIP_EINVAL; }
  for (size_t i = 0; i < count && n; i++)
    if (!d_cols[i]) { set_error("linear_combination: column %zu is null", i); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(lincomb_workspace_bytes(count, n))) != ZKHIP_OK) return rc;
  return fr_linear_combination_device(d_cols, (const uint32_t*)coeffs, count, n, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s, &sc->args);
}

static int perm_args_ok(const void* const* values, const void* const* sigmas, uint32_t n_columns, uint32_t chunk_len, uint32_t log_n, size_t usable_rows,
                        const uint64_t* beta, const uint64_t* gamma, const uint64_t* delta, const uint64_t* omega, const void* z) {
  if (n_columns == 0) return ZKHIP_OK;
  if (!values || !sigmas || !beta || !gamma || !delta || !omega || !z) { set_error("permutation_products: null pointer"); return ZKHIP_EINVAL; }
  if (chunk_len == 0 || log_n > 28 || usable_rows > ((size_t)1 << log_n)) {
    set_error("permutation_products: bad shape (chunk_len %u, log_n %u, usable_rows %zu)", chunk_len, log_n, usable_rows);
    return ZKHIP_EINVAL;
  }
  for (uint32_t i = 0; i < n_columns; i++)
    if (!values[i] || !sigmas[i]) { set_error("permutation_products: column %u is null", i); return ZKHIP_EINVAL; }
  return ZKHIP_OK;
}

int zkhip_permutation_products_device(const void* const* d_values, const void* const* d_sigmas, uint32_t n_columns, uint32_t chunk_len, uint32_t log_n,
                                      size_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4], const uint64_t omega[4],
                                      void* d_z, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((rc = perm_args_ok(d_values, d_sigmas, n_columns, chunk_len, log_n, usable_rows, beta, gamma, delta, omega, d_z)) != ZKHIP_OK) return rc;
  if (n_columns == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(perm_workspace_bytes(n_columns, chunk_len, log_n))) != ZKHIP_OK) return rc;
  return fr_permutation_products_device(d_values, d_sigmas, n_columns, chunk_len, log_n, usable_rows, (const uint32_t*)beta, (const uint32_t*)gamma,
                                        (const uint32_t*)delta, (const uint32_t*)omega, (uint32_t*)d_z, sc->ws.p, sc->ws.cap, s, &sc->args);
}

int zkhip_permutation_products(const uint64_t* const* values, const uint64_t* const* sigmas, uint32_t n_columns, uint32_t chunk_len, uint32_t log_n,
                               size_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4], const uint64_t omega[4],
                               uint64_t* z) {
  ZK_API_RANGE();
  int rc;
  if ((rc = perm_args_ok((const void* const*)values, (const void* const*)sigmas, n_columns, chunk_len, log_n, usable_rows, beta, gamma, delta, omega, z)) != ZKHIP_OK)
    return rc;
  if (n_columns == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  hipStream_t s = H.s;
  const size_t n = (size_t)1 << log_n, bytes = n * 32, nsets = (n_columns + chunk_len - 1) / chunk_len;
  if ((rc = H.sc->poly.reserve((size_t)2 * n_columns * bytes)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly2.reserve(nsets * bytes)) != ZKHIP_OK) return rc;
  std::vector<const void*> d_cols((size_t)2 * n_columns);
  for (uint32_t i = 0; i < n_columns; i++) {
    d_cols[i] = (char*)H.sc->poly.p + (size_t)i * bytes;
    d_cols[n_columns + i] = (char*)H.sc->poly.p + (size_t)(n_columns + i) * bytes;
    HIPCHK(hipMemcpyAsync((void*)d_cols[i], values[i], bytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync((void*)d_cols[n_columns + i], sigmas[i], bytes, hipMemcpyHostToDevice, s));
  }
  if ((rc = zkhip_permutation_products_device(d_cols.data(), d_cols.data() + n_columns, n_columns, chunk_len, log_n, usable_rows, beta, gamma, delta, omega,
                                              H.sc->poly2.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(z, H.sc->poly2.p, nsets * bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_profile_enable(int on) {
  guard_t g(g_mu);
  g_prof_mode = on == 2 ? 2 : (on != 0 ? 1 : 0);
  g_prof_n = 0;
  g_prof_base = 0;
  g_prof_ncalls = 0;
  g_prof_full = false;
  return ZKHIP_OK;
}

int zkhip_profile_read(double* ms, char (*names)[64], int max) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  if (!g_prof_ev_ready || g_prof_n == 0 || g_prof_mode == 2) return 0;
  if (hipEventSynchronize(g_prof_ev[g_prof_n]) != hipSuccess) { set_error("profile_read: event sync failed"); return ZKHIP_EHIP; }
  for (int i = 0; i < g_prof_n && i < max; i++) {
    float t = 0;
    (void)hipEventElapsedTime(&t, g_prof_ev[i], g_prof_ev[i + 1]);
    if (ms) ms[i] = t;
    if (names) snprintf(names[i], 64, "%s", g_prof_names[i]);
  }
  return g_prof_n;
}

int zkhip_profile_read_calls(double* ms, char (*names)[64], int* call_of, int max) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  if (!g_prof_ev_ready || g_prof_mode != 2 || g_prof_ncalls == 0) return 0;
  if (!g_prof_full) g_prof_calls[g_prof_ncalls - 1][1] = g_prof_n;
  int out = 0;
  for (int c = 0; c < g_prof_ncalls; c++) {
    const int base = g_prof_calls[c][0], n = g_prof_calls[c][1];
    if (n == 0) continue;
    if (hipEventSynchronize(g_prof_ev[base + n]) != hipSuccess) { set_error("profile_read_calls: event sync failed"); return ZKHIP_EHIP; }
    for (int i = 0; i < n && out < max; i++, out++) {
      float t = 0;
      (void)hipEventElapsedTime(&t, g_prof_ev[base + i], g_prof_ev[base + i + 1]);
      if (ms) ms[out] = t;
      if (names) snprintf(names[out], 64, "%s", g_prof_names[base + i]);
      if (call_of) call_of[out] = c;
    }
  }
  return out;
}

int zkhip_g1_fixed_base_mul_device(const void* d_scalars, size_t n, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_scalars || !d_out)) { set_error("fixed_base_mul: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  device_ctx& P = primary();
  if ((rc = sc->ws.reserve(g1_fixed_base_workspace(n))) != ZKHIP_OK) return rc;
  if (!P.fixed_table_ready) {              // one-time: 16 windows x 2^15 multiples of the generator (32 MiB), shared by every stream
    if ((rc = P.fixed_table.reserve(g1_fixed_base_table_bytes())) != ZKHIP_OK) return rc;
    if ((rc = g1_fixed_base_table_build((uint32_t*)P.fixed_table.p, sc->ws.p, sc->ws.cap, s)) != ZKHIP_OK) return rc;
    HIPCHK(hipStreamSynchronize(s));
    P.fixed_table_ready = true;
  }
  return g1_fixed_base_mul_device((const uint32_t*)d_scalars, n, (const uint32_t*)P.fixed_table.p, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s);
}

int zkhip_g1_fft_device(void* d_points_xyz, const uint64_t omega[4], uint32_t log_n, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_points_xyz || !omega) { set_error("g1_fft: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 26) { set_error("g1_fft: log_n = %u out of range", log_n); return ZKHIP_EINVAL; }
  const size_t n = (size_t)1 << log_n;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(g1_fft_workspace(n))) != ZKHIP_OK) return rc;
  return g1_fft_device((const uint32_t*)d_points_xyz, 1, (uint32_t*)d_points_xyz, 1, log_n, (const uint32_t*)omega, nullptr, sc->ws.p, sc->ws.cap, s);
}

int zkhip_g_to_lagrange_device(const void* d_g, uint32_t k, void* d_g_lagrange, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_g || !d_g_lagrange) { set_error("g_to_lagrange: null pointer"); return ZKHIP_EINVAL; }
  if (d_g == d_g_lagrange) { set_error("g_to_lagrange: the arrays may not alias"); return ZKHIP_EINVAL; }
  if (k > 26) { set_error("g_to_lagrange: k = %u out of range", k); return ZKHIP_EINVAL; }
  const size_t n = (size_t)1 << k;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(g1_fft_workspace(n))) != ZKHIP_OK) return rc;
  namespace H = zkhip::halo2;
  H::Fr omega = H::fr_root_of_unity();
  for (uint32_t i = k; i < 28; i++) omega = H::detail::mul(omega, omega);
  const H::Fr omega_inv = H::detail::invert(omega), n_inv = H::detail::invert(H::detail::from_u64((uint64_t)n));
  return g1_fft_device((const uint32_t*)d_g, 0, (uint32_t*)d_g_lagrange, 0, k, (const uint32_t*)omega_inv.l, (const uint32_t*)n_inv.l, sc->ws.p, sc->ws.cap, s);
}

// host-buffer form, with the memory of the reference's `g_to_lagrange(g_projective: Vec<C::Curve>, k) -> Vec<C>`: 2^k Jacobian points in, 2^k affine
// points out (the whole function: inverse FFT over the points, the 1/n scaling and Curve::batch_normalize in one call)
int zkhip_g_to_lagrange(const uint64_t* g_xyz, uint32_t k, uint64_t* g_lagrange) {
  ZK_API_RANGE();
  if (!g_xyz || !g_lagrange) { set_error("g_to_lagrange: null pointer"); return ZKHIP_EINVAL; }
  if (k > 26) { set_error("g_to_lagrange: k = %u out of range", k); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  const size_t n = (size_t)1 << k;
  hipStream_t s = H.s;
  scratch* sc = H.sc;
  if ((rc = sc->bases.reserve(n * 96)) != ZKHIP_OK) return rc;
  if ((rc = sc->poly.reserve(n * 64)) != ZKHIP_OK) return rc;
  if ((rc = sc->ws.reserve(g1_fft_workspace(n))) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(sc->bases.p, g_xyz, n * 96, hipMemcpyHostToDevice, s));
  namespace Hh = zkhip::halo2;
  Hh::Fr omega = Hh::fr_root_of_unity();
  for (uint32_t i = k; i < 28; i++) omega = Hh::detail::mul(omega, omega);
  const Hh::Fr omega_inv = Hh::detail::invert(omega), n_inv = Hh::detail::invert(Hh::detail::from_u64((uint64_t)n));
  if ((rc = g1_fft_device((const uint32_t*)sc->bases.p, 1, (uint32_t*)sc->poly.p, 0, k, (const uint32_t*)omega_inv.l, (const uint32_t*)n_inv.l, sc->ws.p, sc->ws.cap, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_lagrange, sc->poly.p, n * 64, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_g1_gen_walk_device(const uint64_t t0[4], const uint64_t d[4], size_t n, void* d_out, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!t0 || !d || (n && !d_out)) { set_error("gen_walk: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(g1_gen_walk_workspace(n))) != ZKHIP_OK) return rc;
  return g1_gen_walk_device((const uint32_t*)t0, (const uint32_t*)d, n, (uint32_t*)d_out, sc->ws.p, sc->ws.cap, s);
}

// ---- Curve::batch_normalize: Jacobian commitments -> affine (what create_proof writes into the transcript) ------------------------
int zkhip_g1_batch_normalize_device(const void* d_points_xyz, size_t n, void* d_out_affine, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_points_xyz || !d_out_affine)) { set_error("batch_normalize: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(g1_batch_normalize_workspace(n))) != ZKHIP_OK) return rc;
  return g1_batch_normalize_device((const uint32_t*)d_points_xyz, n, (uint32_t*)d_out_affine, sc->ws.p, sc->ws.cap, s);
}

int zkhip_g1_batch_normalize(const uint64_t* points_xyz, size_t n, uint64_t* out_affine) {
  ZK_API_RANGE();
  if (n && (!points_xyz || !out_affine)) { set_error("batch_normalize: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * (96 + 64))) != ZKHIP_OK) return rc;
  if ((rc = H.sc->ws.reserve(g1_batch_normalize_workspace(n))) != ZKHIP_OK) return rc;
  char* d = (char*)H.sc->poly.p;
  HIPCHK(hipMemcpyAsync(d, points_xyz, n * 96, hipMemcpyHostToDevice, s));
  if ((rc = g1_batch_normalize_device((const uint32_t*)d, n, (uint32_t*)(d + n * 96), H.sc->ws.p, H.sc->ws.cap, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_affine, d + n * 96, n * 64, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// ---- G2 MSM: best_multiexp::<G2Affine> -------------------------------------------------------------------------------
int zkhip_msm_g2_device(const void* d_scalars, const void* d_bases, size_t n, void* d_out_xyz, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_out_xyz || (n && (!d_scalars || !d_bases))) { set_error("msm_g2: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->ws.reserve(msm_g2_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return msm_g2_device((const uint32_t*)d_scalars, (const uint32_t*)d_bases, n, (uint32_t*)d_out_xyz, sc->ws.p, sc->ws.cap, s);
}

int zkhip_msm_g2(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xyz[24]) {
  ZK_API_RANGE();
  if (!out_xyz || (n && (!scalars || !bases))) { set_error("msm_g2: null pointer"); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  if (n) {
    if ((rc = H.sc->scalars.reserve(n * 32)) != ZKHIP_OK) return rc;
    if ((rc = H.sc->bases.reserve(n * 128)) != ZKHIP_OK) return rc;
    if ((rc = H.sc->ws.reserve(msm_g2_workspace_bytes(n))) != ZKHIP_OK) return rc;
    HIPCHK(hipMemcpyAsync(H.sc->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(H.sc->bases.p, bases, n * 128, hipMemcpyHostToDevice, s));
  }
  if ((rc = msm_g2_device((const uint32_t*)H.sc->scalars.p, (const uint32_t*)H.sc->bases.p, n, (uint32_t*)H.sc->small.p, H.sc->ws.p, H.sc->ws.cap, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(H.L->pinned, H.sc->small.p, 192, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  memcpy(out_xyz, H.L->pinned, 192);
  return ZKHIP_OK;
}

// ---- SRS / key-file validation: every point canonical and on the curve (RawBytes readers) ------------------------------
int zkhip_g1_check_points_device(const void* d_points, size_t n, uint64_t* first_bad, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!first_bad || (n && !d_points)) { set_error("g1_check_points: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  if ((rc = g1_check_points_device((const uint32_t*)d_points, n, (unsigned long long*)sc->small.p, s)) != ZKHIP_OK) return rc;
  unsigned long long v = 0;
  HIPCHK(hipMemcpyAsync(&v, sc->small.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *first_bad = (uint64_t)v;
  return ZKHIP_OK;
}

int zkhip_g1_check_points(const uint64_t* points, size_t n, uint64_t* first_bad) {
  ZK_API_RANGE();
  if (!first_bad || (n && !points)) { set_error("g1_check_points: null pointer"); return ZKHIP_EINVAL; }
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  *first_bad = n;
  const size_t chunk = (size_t)1 << 24;                  // 1 GiB of points per upload
  if ((rc = H.sc->bases.reserve(std::min(n, chunk) * 64)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t m = std::min(chunk, n - lo);
    HIPCHK(hipMemcpyAsync(H.sc->bases.p, points + lo * 8, m * 64, hipMemcpyHostToDevice, H.s));
    if ((rc = g1_check_points_device((const uint32_t*)H.sc->bases.p, m, (unsigned long long*)H.sc->small.p, H.s)) != ZKHIP_OK) return rc;
    HIPCHK(hipMemcpyAsync(H.L->pinned, H.sc->small.p, 8, hipMemcpyDeviceToHost, H.s));
    HIPCHK(hipStreamSynchronize(H.s));
    const unsigned long long v = *(const unsigned long long*)H.L->pinned;
    if (v < m) { *first_bad = lo + (size_t)v; return ZKHIP_OK; }
  }
  return ZKHIP_OK;
}

// ---- SerdeFormat::Processed: compressed G1 points (serde.hip) -------------------------------------------------------------
int zkhip_g1_compress_device(const void* d_points, size_t n, void* d_out32, int flag_layout, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((n && (!d_points || !d_out32)) || flag_layout < 0 || flag_layout > 1) { set_error("g1_compress: bad argument"); return ZKHIP_EINVAL; }
  return g1_compress_device((const uint32_t*)d_points, n, (uint32_t*)d_out32, flag_layout, caller_stream(stream));
}

int zkhip_g1_decompress_device(const void* d_in32, size_t n, void* d_points, int flag_layout, uint64_t* first_bad, void* stream) {
  ZK_API_RANGE();
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!first_bad || (n && (!d_in32 || !d_points)) || flag_layout < 0 || flag_layout > 1) { set_error("g1_decompress: bad argument"); return ZKHIP_EINVAL; }
  hipStream_t s = caller_stream(stream);
  scratch* sc = scratch_for(primary(), s);
  if ((rc = sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  unsigned long long v = (unsigned long long)n;
  HIPCHK(hipMemcpyAsync(sc->small.p, &v, 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));                     // `v` is a stack variable
  if ((rc = g1_decompress_device((const uint32_t*)d_in32, n, (uint32_t*)d_points, flag_layout, (unsigned long long*)sc->small.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(&v, sc->small.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *first_bad = (uint64_t)v;
  return ZKHIP_OK;
}

int zkhip_g1_compress(const uint64_t* points, size_t n, uint8_t* out32, int flag_layout) {
  ZK_API_RANGE();
  if ((n && (!points || !out32)) || flag_layout < 0 || flag_layout > 1) { set_error("g1_compress: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  const size_t chunk = (size_t)1 << 23;                // 512 MiB of points per upload
  if ((rc = H.sc->bases.reserve(std::min(n, chunk) * 64)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly.reserve(std::min(n, chunk) * 32)) != ZKHIP_OK) return rc;
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t m = std::min(chunk, n - lo);
    HIPCHK(hipMemcpyAsync(H.sc->bases.p, points + lo * 8, m * 64, hipMemcpyHostToDevice, H.s));
    if ((rc = g1_compress_device((const uint32_t*)H.sc->bases.p, m, (uint32_t*)H.sc->poly.p, flag_layout, H.s)) != ZKHIP_OK) return rc;
    HIPCHK(hipMemcpyAsync(out32 + lo * 32, H.sc->poly.p, m * 32, hipMemcpyDeviceToHost, H.s));
    HIPCHK(hipStreamSynchronize(H.s));
  }
  return ZKHIP_OK;
}

int zkhip_g1_decompress(const uint8_t* in32, size_t n, uint64_t* points, int flag_layout, uint64_t* first_bad) {
  ZK_API_RANGE();
  if (!first_bad || (n && (!in32 || !points)) || flag_layout < 0 || flag_layout > 1) { set_error("g1_decompress: bad argument"); return ZKHIP_EINVAL; }
  *first_bad = n;
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  const size_t chunk = (size_t)1 << 23;
  if ((rc = H.sc->bases.reserve(std::min(n, chunk) * 64)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->poly.reserve(std::min(n, chunk) * 32)) != ZKHIP_OK) return rc;
  if ((rc = H.sc->small.reserve(4096)) != ZKHIP_OK) return rc;
  for (size_t lo = 0; lo < n; lo += chunk) {
    const size_t m = std::min(chunk, n - lo);
    *(unsigned long long*)H.L->pinned = (unsigned long long)m;
    HIPCHK(hipMemcpyAsync(H.sc->small.p, H.L->pinned, 8, hipMemcpyHostToDevice, H.s));
    HIPCHK(hipMemcpyAsync(H.sc->poly.p, in32 + lo * 32, m * 32, hipMemcpyHostToDevice, H.s));
    if ((rc = g1_decompress_device((const uint32_t*)H.sc->poly.p, m, (uint32_t*)H.sc->bases.p, flag_layout, (unsigned long long*)H.sc->small.p, H.s)) != ZKHIP_OK) return rc;
    HIPCHK(hipMemcpyAsync(points + lo * 8, H.sc->bases.p, m * 64, hipMemcpyDeviceToHost, H.s));
    HIPCHK(hipMemcpyAsync(H.L->pinned, H.sc->small.p, 8, hipMemcpyDeviceToHost, H.s));
    HIPCHK(hipStreamSynchronize(H.s));
    const unsigned long long v = *(const unsigned long long*)H.L->pinned;
    if (v < m) { *first_bad = lo + (size_t)v; return ZKHIP_OK; }
  }
  return ZKHIP_OK;
}

// ---- parity hooks ----------------------------------------------------------------------------------
int zkhip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  ZK_API_RANGE();
  if (field < 0 || field > 1 || op < 0 || op > 3 || (n && (!a || !b || !out))) { set_error("test_field_op: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * 96)) != ZKHIP_OK) return rc;
  char* d = (char*)H.sc->poly.p;
  HIPCHK(hipMemcpyAsync(d, a, n * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + n * 32, b, n * 32, hipMemcpyHostToDevice, s));
  if ((rc = test_field_op(field, op, (uint32_t*)d, (uint32_t*)(d + n * 32), (uint32_t*)(d + n * 64), n, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out, d + n * 64, n * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_test_g1_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out_xyz, size_t n) {
  ZK_API_RANGE();
  if (op < 0 || op > 4 || (n && (!a || !b || !out_xyz))) { set_error("test_g1_op: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * (64 + 64 + 96))) != ZKHIP_OK) return rc;
  char* d = (char*)H.sc->poly.p;
  HIPCHK(hipMemcpyAsync(d, a, n * 64, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + n * 64, b, n * 64, hipMemcpyHostToDevice, s));
  if ((rc = test_g1_op(op, (uint32_t*)d, (uint32_t*)(d + n * 64), (uint32_t*)(d + n * 128), n, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, d + n * 128, n * 96, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_test_g2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out_xyz, size_t n) {
  ZK_API_RANGE();
  if (op < 0 || op > 6 || (n && (!a || !b || !out_xyz))) { set_error("test_g2_op: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  lane_hold H;
  if (H.rc != ZKHIP_OK) return H.rc;
  int rc;
  hipStream_t s = H.s;
  if ((rc = H.sc->poly.reserve(n * (128 + 128 + 192))) != ZKHIP_OK) return rc;
  char* d = (char*)H.sc->poly.p;
  HIPCHK(hipMemcpyAsync(d, a, n * 128, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d + n * 128, b, n * 128, hipMemcpyHostToDevice, s));
  if ((rc = test_g2_op(op, (uint32_t*)d, (uint32_t*)(d + n * 128), (uint32_t*)(d + n * 256), n, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, d + n * 256, n * 192, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

}  // extern "C"
