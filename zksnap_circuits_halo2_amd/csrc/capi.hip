// C ABI of libzkhip.so (include/zkhip.h): context, device buffers, registered-base residency, and the host-buffer
// wrappers around the device paths in msm.hip / ntt.hip.  No CPU arithmetic lives here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include <mutex>
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

// ---- profiler -------------------------------------------------------------------------------------
static bool g_prof_on = false;
static const int PROF_MAX = 32;
static hipEvent_t g_prof_ev[PROF_MAX + 1];
static bool g_prof_ev_ready = false;
static int g_prof_n = 0;   // number of marks recorded after the begin event
static char g_prof_names[PROF_MAX][64];

void prof_begin(hipStream_t stream) {
  if (!g_prof_on) return;
  if (!g_prof_ev_ready) {
    for (int i = 0; i <= PROF_MAX; i++) (void)hipEventCreate(&g_prof_ev[i]);
    g_prof_ev_ready = true;
  }
  g_prof_n = 0;
  (void)hipEventRecord(g_prof_ev[0], stream);
}

void prof_mark(hipStream_t stream, const char* name) {
  if (!g_prof_on || !g_prof_ev_ready || g_prof_n >= PROF_MAX) return;
  snprintf(g_prof_names[g_prof_n], 64, "%s", name);
  g_prof_n++;
  (void)hipEventRecord(g_prof_ev[g_prof_n], stream);
}


#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

struct dev_buf {       // grow-only device scratch
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return ZKHIP_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8;
    if (hipMalloc(&p, want) != hipSuccess) {
      if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc(%zu) failed", bytes); p = nullptr; return ZKHIP_ENOMEM; }
      want = bytes;
    }
    cap = want;
    return ZKHIP_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct context {
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  dev_buf ws, scalars, bases, poly, poly2, small, ntt_tmp, vm, fixed_table;
  bool fixed_table_ready = false;   // multiples of the generator for zkhip_g1_fixed_base_mul_device
  std::map<const void*, prepared_bases*> registered;   // host ptr -> prepared table (slice 0 of the table = the bases themselves)
  std::map<uint64_t, prepared_bases*> handles;          // zkhip_prepare_bases_device handles
  uint64_t next_handle = 1;
};

static std::recursive_mutex g_mu;
static context g_ctx;

static int ensure_init() {
  if (g_ctx.ready) return ZKHIP_OK;
  return zkhip_init(nullptr, 0);
}

// prepared table + point offset for `bases` if it lies inside a registered range with room for n points
static const prepared_bases* find_registered(const uint64_t* bases, size_t n, size_t* off) {
  for (auto& kv : g_ctx.registered) {
    const char* lo = (const char*)kv.first;
    const char* hi = lo + kv.second->n * 64;
    const char* q = (const char*)bases;
    if (q >= lo && q + n * 64 <= hi && ((q - lo) % 64) == 0) { *off = (size_t)(q - lo) / 64; return kv.second; }
  }
  return nullptr;
}

// `stream` argument of the `_device` entry points: NULL is HIP's default (legacy) stream -- stream 0, which is what a caller
// holding "the default stream" (e.g. torch.cuda.current_stream().cuda_stream == 0) passes, and which is ordered with that caller's
// other default-stream work.  (The host-buffer entry points use the library's own non-blocking stream and synchronise it.)
static inline hipStream_t caller_stream(void* stream) { return (hipStream_t)stream; }

}  // namespace zkhip

using namespace zkhip;
typedef std::lock_guard<std::recursive_mutex> guard_t;

extern "C" {

int zkhip_init(const int* devices, int ndev) {
  guard_t g(g_mu);
  if (ndev > 1) { set_error("zkhip_init: one process drives one GPU (ndev = %d)", ndev); return ZKHIP_EINVAL; }
  int dev = 0;
  if (devices && ndev == 1) dev = devices[0];
  else if (const char* e = getenv("ZKHIP_DEVICE")) dev = atoi(e);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { set_error("no HIP device available"); return ZKHIP_ENODEV; }
  if (dev < 0 || dev >= count) { set_error("device %d out of range (%d devices)", dev, count); return ZKHIP_EINVAL; }
  if (g_ctx.ready && g_ctx.device == dev) return ZKHIP_OK;
  if (g_ctx.ready) zkhip_shutdown();
  if (hipSetDevice(dev) != hipSuccess) { set_error("hipSetDevice(%d) failed", dev); return ZKHIP_ENODEV; }
  if (hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return ZKHIP_ENODEV; }
  g_ctx.device = dev;
  g_ctx.ready = true;
  return ZKHIP_OK;
}

void zkhip_shutdown(void) {
  guard_t g(g_mu);
  if (!g_ctx.ready) return;
  (void)hipSetDevice(g_ctx.device);
  (void)hipStreamSynchronize(g_ctx.stream);
  ntt_clear_cache();
  for (auto& kv : g_ctx.registered) release_prepared(kv.second);
  g_ctx.registered.clear();
  for (auto& kv : g_ctx.handles) release_prepared(kv.second);
  g_ctx.handles.clear();
  g_ctx.ws.release(); g_ctx.scalars.release(); g_ctx.bases.release(); g_ctx.poly.release(); g_ctx.poly2.release(); g_ctx.small.release(); g_ctx.ntt_tmp.release(); g_ctx.vm.release(); g_ctx.fixed_table.release(); g_ctx.fixed_table_ready = false;
  (void)hipStreamDestroy(g_ctx.stream);
  g_ctx.stream = nullptr;
  g_ctx.ready = false;
}

const char* zkhip_last_error(void) { return g_err; }

int zkhip_device_name(char* buf, size_t len) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, g_ctx.device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return ZKHIP_OK;
}

int zkhip_msm_window_bits(size_t n) { return msm_pick_window(n); }

// ---- MSM -----------------------------------------------------------------------------------------
int zkhip_msm_g1_device_c(const void* d_scalars, const void* d_bases, size_t n, void* d_out_xyz, int window_bits, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_out_xyz || (n && (!d_scalars || !d_bases))) { set_error("msm: null pointer"); return ZKHIP_EINVAL; }
  const int c = window_bits > 0 ? window_bits : msm_pick_window(n ? n : 1);
  const size_t need = n ? msm_workspace_bytes(n, c) : 0;
  if ((rc = g_ctx.ws.reserve(need)) != ZKHIP_OK) return rc;
  hipStream_t s = caller_stream(stream);
  return msm_g1_device((const uint32_t*)d_scalars, (const uint32_t*)d_bases, n, (uint32_t*)d_out_xyz, g_ctx.ws.p, g_ctx.ws.cap, c, s);
}

int zkhip_msm_g1_device(const void* d_scalars, const void* d_bases, size_t n, void* d_out_xyz, void* stream) {
  return zkhip_msm_g1_device_c(d_scalars, d_bases, n, d_out_xyz, 0, stream);
}

int zkhip_msm_g1(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xyz[12]) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!out_xyz || (n && (!scalars || !bases))) { set_error("msm: null pointer"); return ZKHIP_EINVAL; }
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.small.reserve(4096)) != ZKHIP_OK) return rc;
  const uint32_t* d_bases = nullptr;
  const prepared_bases* pb = nullptr;
  size_t off = 0;
  if (n) {
    if ((rc = g_ctx.scalars.reserve(n * 32)) != ZKHIP_OK) return rc;
    HIPCHK(hipMemcpyAsync(g_ctx.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, s));
    pb = find_registered(bases, n, &off);
    if (!pb) {
      if ((rc = g_ctx.bases.reserve(n * 64)) != ZKHIP_OK) return rc;
      HIPCHK(hipMemcpyAsync(g_ctx.bases.p, bases, n * 64, hipMemcpyHostToDevice, s));
      d_bases = (const uint32_t*)g_ctx.bases.p;
    }
  }
  if (pb) {
    if ((rc = g_ctx.ws.reserve(msm_workspace_bytes(n, pb->c, true))) != ZKHIP_OK) return rc;
    rc = msm_g1_device((const uint32_t*)g_ctx.scalars.p, nullptr, n, (uint32_t*)g_ctx.small.p, g_ctx.ws.p, g_ctx.ws.cap, 0, s, pb, off);
  } else {
    rc = zkhip_msm_g1_device_c(g_ctx.scalars.p, d_bases, n, g_ctx.small.p, 0, s);
  }
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, g_ctx.small.p, 96, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// `batch` scalar vectors (contiguous, n elements each) against the same bases; out: batch Jacobian points.
// Registered bases: one batched launch set; otherwise one general-path MSM per vector.
int zkhip_msm_g1_batch(const uint64_t* scalars, const uint64_t* bases, size_t n, size_t batch, uint64_t* out_xyz) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!out_xyz || (n && batch && (!scalars || !bases))) { set_error("msm_batch: null pointer"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  size_t off = 0;
  const prepared_bases* pb = n ? find_registered(bases, n, &off) : nullptr;
  if (!pb || pb->c > 16) {
    for (size_t k = 0; k < batch; k++)
      if ((rc = zkhip_msm_g1(scalars + k * n * 4, bases, n, out_xyz + k * 12)) != ZKHIP_OK) return rc;
    return ZKHIP_OK;
  }
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.scalars.reserve(n * batch * 32)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.small.reserve(4096 + batch * 96)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.scalars.p, scalars, n * batch * 32, hipMemcpyHostToDevice, s));
  // reuse the handle-based entry point through a temporary handle for the registered table
  const uint64_t tmp_handle = g_ctx.next_handle++;
  g_ctx.handles[tmp_handle] = const_cast<prepared_bases*>(pb);
  rc = zkhip_msm_g1_prepared_batch_device(tmp_handle, off, g_ctx.scalars.p, n, batch, n, g_ctx.small.p, s);
  g_ctx.handles.erase(tmp_handle);
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, g_ctx.small.p, batch * 96, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_register_bases(const uint64_t* bases, size_t n) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!bases || n == 0) { set_error("register_bases: empty"); return ZKHIP_EINVAL; }
  if (g_ctx.registered.count(bases)) zkhip_unregister_bases(bases);
  if ((rc = g_ctx.bases.reserve(n * 64)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.bases.p, bases, n * 64, hipMemcpyHostToDevice, g_ctx.stream));
  prepared_bases* pb = nullptr;
  if ((rc = prepare_bases_device((const uint32_t*)g_ctx.bases.p, n, g_ctx.stream, &pb)) != ZKHIP_OK) return rc;
  g_ctx.registered[bases] = pb;
  return ZKHIP_OK;
}

int zkhip_prepare_bases_device(const void* d_bases, size_t n, uint64_t* handle) {
  return zkhip_prepare_bases_device_c(d_bases, n, 0, handle);
}

int zkhip_prepare_bases_device_c(const void* d_bases, size_t n, int window_bits, uint64_t* handle) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_bases || !handle || n == 0) { set_error("prepare_bases: bad argument"); return ZKHIP_EINVAL; }
  prepared_bases* pb = nullptr;
  HIPCHK(hipDeviceSynchronize());           // d_bases may still be being written on the caller's stream; one-time call
  if ((rc = prepare_bases_device((const uint32_t*)d_bases, n, g_ctx.stream, &pb, window_bits)) != ZKHIP_OK) return rc;
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
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("msm_prepared: unknown handle"); return ZKHIP_EINVAL; }
  if (!d_out_xyz || (n && !d_scalars)) { set_error("msm_prepared: null pointer"); return ZKHIP_EINVAL; }
  const prepared_bases* pb = it->second;
  if ((rc = g_ctx.ws.reserve(n ? msm_workspace_bytes(n, pb->c, true) : 0)) != ZKHIP_OK) return rc;
  return msm_g1_device((const uint32_t*)d_scalars, nullptr, n, (uint32_t*)d_out_xyz, g_ctx.ws.p, g_ctx.ws.cap, 0,
                       caller_stream(stream), pb, offset);
}

int zkhip_unregister_bases(const uint64_t* bases) {
  guard_t g(g_mu);
  auto it = g_ctx.registered.find(bases);
  if (it == g_ctx.registered.end()) { set_error("unregister_bases: pointer not registered"); return ZKHIP_EINVAL; }
  (void)hipDeviceSynchronize();             // the table may still be in use on a caller's stream
  release_prepared(it->second);
  g_ctx.registered.erase(it);
  return ZKHIP_OK;
}

int zkhip_msm_g1_prepared_batch_device(uint64_t handle, size_t offset, const void* d_scalars, size_t n, size_t batch, size_t scalar_stride,
                                       void* d_out_xyz, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  auto it = g_ctx.handles.find(handle);
  if (it == g_ctx.handles.end()) { set_error("msm_prepared_batch: unknown handle"); return ZKHIP_EINVAL; }
  if (!d_out_xyz || (n && batch && !d_scalars) || (batch > 1 && scalar_stride < n)) { set_error("msm_prepared_batch: bad argument"); return ZKHIP_EINVAL; }
  const prepared_bases* pb = it->second;
  hipStream_t s = caller_stream(stream);
  // tables built for wide windows (n >= 2^20) do not batch: those MSMs are throughput-bound one at a time
  size_t group = (pb->c > 16 || batch <= 1) ? 1 : batch;
  while (group > 1 && ((size_t)((256 + pb->c - 1) / pb->c) * n * group >= (1ull << 31) || (group << (pb->c - 1)) > (1ull << 22))) group = (group + 1) / 2;   // <= 4 Mi buckets per launch set
  for (size_t k0 = 0; k0 < batch; k0 += group) {
    const size_t kk = batch - k0 < group ? batch - k0 : group;
    if ((rc = g_ctx.ws.reserve(n ? msm_workspace_bytes(n, pb->c, true, kk) : 0)) != ZKHIP_OK) return rc;
    rc = msm_g1_device((const uint32_t*)d_scalars + k0 * scalar_stride * 8, nullptr, n, (uint32_t*)d_out_xyz + k0 * 24, g_ctx.ws.p, g_ctx.ws.cap, 0, s, pb,
                       offset, kk, scalar_stride);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

int zkhip_g1_sum_device(const void* d_points_xyz, int m, void* d_out_xyz, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (m < 0 || !d_out_xyz || (m && !d_points_xyz)) { set_error("g1_sum: bad argument"); return ZKHIP_EINVAL; }
  return sum_jacobian_device((const uint32_t*)d_points_xyz, m, (uint32_t*)d_out_xyz, caller_stream(stream));
}

int zkhip_g1_sum(const uint64_t* points_xyz, int m, uint64_t out_xyz[12]) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (m < 0 || !out_xyz || (m && !points_xyz)) { set_error("g1_sum: bad argument"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.small.reserve(4096 + (size_t)m * 96)) != ZKHIP_OK) return rc;
  char* d = (char*)g_ctx.small.p;
  if (m) HIPCHK(hipMemcpyAsync(d + 4096, points_xyz, (size_t)m * 96, hipMemcpyHostToDevice, g_ctx.stream));
  if ((rc = sum_jacobian_device((const uint32_t*)(d + 4096), m, (uint32_t*)d, g_ctx.stream)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz, d, 96, hipMemcpyDeviceToHost, g_ctx.stream));
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  return ZKHIP_OK;
}

// ---- NTT / domain ----------------------------------------------------------------------------------
// runs ntt_transform with scratch from the context (two buffers of batch * 2^L elements when the transform is multi-pass)
static int run_transform(const uint32_t* d_in, uint32_t in_len, uint32_t in_stride, uint32_t* d_out, uint32_t out_len, uint32_t out_stride,
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
    int rc = g_ctx.ntt_tmp.reserve(one * 32 * (np >= 3 ? 2 : 1));
    if (rc != ZKHIP_OK) return rc;
    t0 = (uint32_t*)g_ctx.ntt_tmp.p;
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

int zkhip_ntt_fr_batch_device(void* d_a, const uint64_t omega[4], uint32_t log_n, uint32_t batch, size_t stride, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega || log_n > 28 || stride < ((size_t)1 << log_n) || stride >= ((size_t)1 << 32)) { set_error("ntt_batch: bad argument"); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return run_transform((const uint32_t*)d_a, N, (uint32_t)stride, (uint32_t*)d_a, N, (uint32_t)stride, batch, log_n, (const uint32_t*)omega, nullptr, 0,
                       nullptr, 0, caller_stream(stream));
}

int zkhip_ifft_scaled_batch_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], uint32_t batch, size_t stride,
                                   void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega_inv || !divisor || log_n > 28 || stride < ((size_t)1 << log_n) || stride >= ((size_t)1 << 32)) { set_error("ifft_batch: bad argument"); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return run_transform((const uint32_t*)d_a, N, (uint32_t)stride, (uint32_t*)d_a, N, (uint32_t)stride, batch, log_n, (const uint32_t*)omega_inv, nullptr, 0,
                       (const uint32_t*)divisor, 1, caller_stream(stream));
}

int zkhip_ntt_fr_device(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega) { set_error("ntt: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 28) { set_error("ntt: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return run_transform((const uint32_t*)d_a, N, N, (uint32_t*)d_a, N, N, 1, log_n, (const uint32_t*)omega, nullptr, 0, nullptr, 0, caller_stream(stream));
}

int zkhip_ifft_scaled_device(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !omega_inv || !divisor) { set_error("ifft: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 28) { set_error("ifft: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  const uint32_t N = 1u << log_n;
  return run_transform((const uint32_t*)d_a, N, N, (uint32_t*)d_a, N, N, 1, log_n, (const uint32_t*)omega_inv, nullptr, 0, (const uint32_t*)divisor, 1,
                       caller_stream(stream));
}

int zkhip_mul_periodic_device(void* d_a, size_t n, const void* d_table, uint32_t period, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((n && !d_a) || !d_table) { set_error("mul_periodic: null pointer"); return ZKHIP_EINVAL; }
  return fr_mul_periodic_device((uint32_t*)d_a, n, (const uint32_t*)d_table, period, caller_stream(stream));
}

static int host_transform(const uint64_t* in, size_t in_len, uint64_t* out, size_t out_len, uint32_t log_n, const uint64_t* omega,
                          const uint32_t* in_scale, uint32_t in_period, const uint32_t* out_scale, uint32_t out_period) {
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (log_n > 28) { set_error("ntt: log_n = %u > 28", log_n); return ZKHIP_EINVAL; }
  if (!in || !out || !omega) { set_error("ntt: null pointer"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << log_n;
  if (in_len > N || out_len > N) { set_error("ntt: length exceeds domain"); return ZKHIP_EINVAL; }
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve(N * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.poly.p, in, in_len * 32, hipMemcpyHostToDevice, s));
  rc = run_transform((const uint32_t*)g_ctx.poly.p, (uint32_t)in_len, (uint32_t)N, (uint32_t*)g_ctx.poly.p, (uint32_t)out_len, (uint32_t)N, 1, log_n,
                     (const uint32_t*)omega, in_scale, in_period, out_scale, out_period, s);
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out, g_ctx.poly.p, out_len * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// `batch` contiguous polynomials of 2^log_n elements, transformed in place in one launch set
int zkhip_ntt_fr_batch(uint64_t* a, const uint64_t omega[4], uint32_t log_n, uint32_t batch) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!a || !omega || log_n > 28) { set_error("ntt_batch: bad argument"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  const size_t N = (size_t)1 << log_n, total = N * batch;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve(total * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.poly.p, a, total * 32, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_ntt_fr_batch_device(g_ctx.poly.p, omega, log_n, batch, N, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(a, g_ctx.poly.p, total * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_ntt_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
  guard_t g(g_mu);
  const size_t N = (size_t)1 << (log_n > 28 ? 0 : log_n);
  return host_transform(a, N, a, N, log_n, omega, nullptr, 0, nullptr, 0);
}

int zkhip_ifft_scaled(uint64_t* a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]) {
  guard_t g(g_mu);
  if (!divisor) { set_error("ifft: null divisor"); return ZKHIP_EINVAL; }
  const size_t N = (size_t)1 << (log_n > 28 ? 0 : log_n);
  return host_transform(a, N, a, N, log_n, omega_inv, nullptr, 0, (const uint32_t*)divisor, 1);
}

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

int zkhip_coeff_to_extended(const uint64_t* a, uint32_t k, uint64_t* out, uint32_t ext_k, const uint64_t ext_omega[4], const uint64_t zeta[4]) {
  guard_t g(g_mu);
  if (!a || !out || !ext_omega || !zeta || k > ext_k || ext_k > 28) { set_error("coeff_to_extended: bad argument"); return ZKHIP_EINVAL; }
  uint32_t sc[24];
  coset_scales(zeta, nullptr, sc);
  return host_transform(a, (size_t)1 << k, out, (size_t)1 << ext_k, ext_k, ext_omega, sc, 3, nullptr, 0);
}

int zkhip_extended_to_coeff(uint64_t* a, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                            const uint64_t zeta[4], uint64_t* out, size_t out_len) {
  guard_t g(g_mu);
  if (!a || !out || !ext_omega_inv || !ext_divisor || !zeta || ext_k > 28) { set_error("extended_to_coeff: bad argument"); return ZKHIP_EINVAL; }
  uint32_t sc[24];
  coset_scales(zeta, ext_divisor, sc);
  return host_transform(a, (size_t)1 << ext_k, out, out_len, ext_k, ext_omega_inv, nullptr, 0, sc, 3);
}

// device-resident forms, `batch` polynomials per launch set (polynomial b at base + b * stride elements)
int zkhip_coeff_to_extended_device(const void* d_a, size_t a_stride, uint32_t k, void* d_out, size_t out_stride, uint32_t ext_k, uint32_t batch,
                                   const uint64_t ext_omega[4], const uint64_t zeta[4], void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !d_out || !ext_omega || !zeta || k > ext_k || ext_k > 28 || a_stride < ((size_t)1 << k) || out_stride < ((size_t)1 << ext_k) ||
      a_stride >= ((size_t)1 << 32) || out_stride >= ((size_t)1 << 32)) { set_error("coeff_to_extended: bad argument"); return ZKHIP_EINVAL; }
  if (batch == 0) return ZKHIP_OK;
  uint32_t sc[24];
  coset_scales(zeta, nullptr, sc);
  return run_transform((const uint32_t*)d_a, 1u << k, (uint32_t)a_stride, (uint32_t*)d_out, 1u << ext_k, (uint32_t)out_stride, batch, ext_k,
                       (const uint32_t*)ext_omega, sc, 3, nullptr, 0, caller_stream(stream));
}

int zkhip_extended_to_coeff_device(const void* d_a, size_t a_stride, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                                   const uint64_t zeta[4], void* d_out, size_t out_stride, size_t out_len, uint32_t batch, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_a || !d_out || !ext_omega_inv || !ext_divisor || !zeta || ext_k > 28 || out_len > ((size_t)1 << ext_k) || a_stride < ((size_t)1 << ext_k) ||
      out_stride < out_len || a_stride >= ((size_t)1 << 32) || out_stride >= ((size_t)1 << 32)) { set_error("extended_to_coeff: bad argument"); return ZKHIP_EINVAL; }
  if (batch == 0 || out_len == 0) return ZKHIP_OK;
  uint32_t sc[24];
  coset_scales(zeta, ext_divisor, sc);
  return run_transform((const uint32_t*)d_a, 1u << ext_k, (uint32_t)a_stride, (uint32_t*)d_out, (uint32_t)out_len, (uint32_t)out_stride, batch, ext_k,
                       (const uint32_t*)ext_omega_inv, nullptr, 0, sc, 3, caller_stream(stream));
}

int zkhip_mul_periodic(uint64_t* a, size_t n, const uint64_t* table, uint32_t period) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((n && !a) || !table || period == 0) { set_error("mul_periodic: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve(n * 32)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.poly2.reserve((size_t)period * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.poly.p, a, n * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g_ctx.poly2.p, table, (size_t)period * 32, hipMemcpyHostToDevice, s));
  if ((rc = fr_mul_periodic_device((uint32_t*)g_ctx.poly.p, n, (const uint32_t*)g_ctx.poly2.p, period, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(a, g_ctx.poly.p, n * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

// ---- row a7: Fr-vector primitives ---------------------------------------------------------------------
int zkhip_fr_eval_polynomial_device(const void* d_poly, size_t n, const uint64_t point[4], void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!point || !d_out || (n && !d_poly)) { set_error("eval_polynomial: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_eval_polynomial_device((const uint32_t*)d_poly, n, (const uint32_t*)point, (uint32_t*)d_out, g_ctx.ws.p, g_ctx.ws.cap,
                                   caller_stream(stream));
}

int zkhip_fr_eval_polynomial_batch_device(const void* const* d_polys, size_t count, size_t n, const uint64_t point[4], void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (count && (!d_polys || !point || !d_out)) { set_error("eval_polynomial_batch: null pointer"); return ZKHIP_EINVAL; }
  for (size_t i = 0; i < count && n; i++)
    if (!d_polys[i]) { set_error("eval_polynomial_batch: polynomial %zu is null", i); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(poly_batch_workspace_bytes(n, count))) != ZKHIP_OK) return rc;
  return fr_eval_polynomial_batch_device(d_polys, count, n, (const uint32_t*)point, (uint32_t*)d_out, g_ctx.ws.p, g_ctx.ws.cap,
                                         caller_stream(stream));
}

int zkhip_fr_kate_division_device(const void* d_a, size_t n, const uint64_t b[4], void* d_q, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!b || (n > 1 && (!d_a || !d_q))) { set_error("kate_division: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_kate_division_device((const uint32_t*)d_a, n, (const uint32_t*)b, (uint32_t*)d_q, g_ctx.ws.p, g_ctx.ws.cap,
                                 caller_stream(stream));
}

int zkhip_fr_batch_invert_device(void* d_a, size_t n, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && !d_a) { set_error("batch_invert: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_batch_invert_device((uint32_t*)d_a, n, g_ctx.ws.p, g_ctx.ws.cap, caller_stream(stream));
}

int zkhip_fr_prefix_product_device(const void* d_v, size_t n, void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_v || !d_out)) { set_error("prefix_product: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(poly_workspace_bytes(n))) != ZKHIP_OK) return rc;
  return fr_prefix_product_device((const uint32_t*)d_v, n, (uint32_t*)d_out, g_ctx.ws.p, g_ctx.ws.cap, caller_stream(stream));
}

// host-buffer wrappers: upload to the poly scratch, run, download
static int host_vec_op(int op, const uint64_t* in, size_t n_in, const uint64_t* c, uint64_t* out, size_t n_out) {
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve((n_in + 1) * 32)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.poly2.reserve((n_out + 1) * 32)) != ZKHIP_OK) return rc;
  if (n_in) HIPCHK(hipMemcpyAsync(g_ctx.poly.p, in, n_in * 32, hipMemcpyHostToDevice, s));
  switch (op) {
    case 0: rc = zkhip_fr_eval_polynomial_device(g_ctx.poly.p, n_in, c, g_ctx.poly2.p, s); break;
    case 1: rc = zkhip_fr_kate_division_device(g_ctx.poly.p, n_in, c, g_ctx.poly2.p, s); break;
    case 2: rc = zkhip_fr_batch_invert_device(g_ctx.poly.p, n_in, s); break;
    default: rc = zkhip_fr_prefix_product_device(g_ctx.poly.p, n_in, g_ctx.poly2.p, s); break;
  }
  if (rc != ZKHIP_OK) return rc;
  if (n_out) HIPCHK(hipMemcpyAsync(out, op == 2 ? g_ctx.poly.p : g_ctx.poly2.p, n_out * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_fr_eval_polynomial(const uint64_t* poly, size_t n, const uint64_t point[4], uint64_t out[4]) {
  guard_t g(g_mu);
  if (!point || !out || (n && !poly)) { set_error("eval_polynomial: null pointer"); return ZKHIP_EINVAL; }
  return host_vec_op(0, poly, n, point, out, 1);
}

int zkhip_fr_kate_division(const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* q) {
  guard_t g(g_mu);
  if (!b || (n > 1 && (!a || !q))) { set_error("kate_division: null pointer"); return ZKHIP_EINVAL; }
  if (n < 2) return ZKHIP_OK;
  return host_vec_op(1, a, n, b, q, n - 1);
}

int zkhip_fr_batch_invert(uint64_t* a, size_t n) {
  guard_t g(g_mu);
  if (n && !a) { set_error("batch_invert: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  return host_vec_op(2, a, n, nullptr, a, n);
}

int zkhip_fr_prefix_product(const uint64_t* v, size_t n, uint64_t* out) {
  guard_t g(g_mu);
  if (n && (!v || !out)) { set_error("prefix_product: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  return host_vec_op(3, v, n, nullptr, out, n);
}

// ---- lookup argument: permute_expression_pair ----------------------------------------------------------------------
int zkhip_lookup_permute_device(const void* d_input, const void* d_table, size_t usable_rows, void* d_permuted_input, void* d_permuted_table,
                                void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (usable_rows && (!d_input || !d_table || !d_permuted_input || !d_permuted_table)) { set_error("lookup_permute: null pointer"); return ZKHIP_EINVAL; }
  if (usable_rows == 0) return ZKHIP_OK;
  if ((rc = g_ctx.vm.reserve(lookup_permute_workspace_bytes(usable_rows))) != ZKHIP_OK) return rc;
  return lookup_permute_device((const uint32_t*)d_input, (const uint32_t*)d_table, usable_rows, (uint32_t*)d_permuted_input,
                               (uint32_t*)d_permuted_table, g_ctx.vm.p, g_ctx.vm.cap, caller_stream(stream));
}

int zkhip_lookup_permute(const uint64_t* input, const uint64_t* table, size_t usable_rows, uint64_t* permuted_input, uint64_t* permuted_table) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (usable_rows && (!input || !table || !permuted_input || !permuted_table)) { set_error("lookup_permute: null pointer"); return ZKHIP_EINVAL; }
  if (usable_rows == 0) return ZKHIP_OK;
  const size_t bytes = usable_rows * 32;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve(2 * bytes)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.poly2.reserve(2 * bytes)) != ZKHIP_OK) return rc;
  char* in = (char*)g_ctx.poly.p;
  char* out = (char*)g_ctx.poly2.p;
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
  guard_t g(g_mu);
  if (!d_ptr) return ZKHIP_OK;
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipDeviceSynchronize());               // kernels queued on any stream may still use it
  HIPCHK(hipFree(d_ptr));
  return ZKHIP_OK;
}

int zkhip_upload(void* d_dst, const void* src, size_t bytes) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (bytes && (!d_dst || !src)) { set_error("upload: null pointer"); return ZKHIP_EINVAL; }
  if (bytes == 0) return ZKHIP_OK;
  HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));      // default stream, blocking
  return ZKHIP_OK;
}

int zkhip_download(void* dst, const void* d_src, size_t bytes) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (bytes && (!dst || !d_src)) { set_error("download: null pointer"); return ZKHIP_EINVAL; }
  if (bytes == 0) return ZKHIP_OK;
  HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return ZKHIP_OK;
}

int zkhip_sync(void) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  HIPCHK(hipDeviceSynchronize());
  return ZKHIP_OK;
}

// ---- section 8(f): row programs and grand products -------------------------------------------------------------
int zkhip_fr_eval_rows_device(const zkhip_vm_program* prog, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows,
                              int accumulate, void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((rc = row_vm_validate(prog, n_columns, log_rows, accumulate)) != ZKHIP_OK) return rc;
  if (!d_out || (n_columns && !d_columns)) { set_error("eval_rows: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.vm.reserve(row_vm_workspace_bytes(prog, n_columns, log_rows))) != ZKHIP_OK) return rc;
  return row_vm_device(prog, d_columns, n_columns, log_rows, accumulate, (uint32_t*)d_out, g_ctx.vm.p, g_ctx.vm.cap,
                       caller_stream(stream));
}

int zkhip_fr_eval_rows(const zkhip_vm_program* prog, const uint64_t* const* columns, uint32_t n_columns, uint32_t log_rows,
                       int accumulate, uint64_t* out) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if ((rc = row_vm_validate(prog, n_columns, log_rows, accumulate)) != ZKHIP_OK) return rc;
  if (!out || (n_columns && !columns)) { set_error("eval_rows: null pointer"); return ZKHIP_EINVAL; }
  const size_t rows = (size_t)1 << log_rows, bytes = rows * 32;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve((size_t)(n_columns ? n_columns : 1) * bytes)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.poly2.reserve(bytes)) != ZKHIP_OK) return rc;
  std::vector<const void*> d_cols(n_columns);
  for (uint32_t i = 0; i < n_columns; i++) {
    if (!columns[i]) { set_error("eval_rows: column %u is null", i); return ZKHIP_EINVAL; }
    d_cols[i] = (char*)g_ctx.poly.p + (size_t)i * bytes;
    HIPCHK(hipMemcpyAsync((void*)d_cols[i], columns[i], bytes, hipMemcpyHostToDevice, s));
  }
  if (accumulate) HIPCHK(hipMemcpyAsync(g_ctx.poly2.p, out, bytes, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_fr_eval_rows_device(prog, d_cols.data(), n_columns, log_rows, accumulate, g_ctx.poly2.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(out, g_ctx.poly2.p, bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_fr_grand_product_device(const void* d_num, void* d_den, size_t n, void* d_z, void* stream) {
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
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!num || !den || !z)) { set_error("grand_product: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = g_ctx.stream;
  if ((rc = g_ctx.poly.reserve(n * 32)) != ZKHIP_OK) return rc;
  if ((rc = g_ctx.poly2.reserve(n * 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(g_ctx.poly.p, num, n * 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g_ctx.poly2.p, den, n * 32, hipMemcpyHostToDevice, s));
  if ((rc = zkhip_fr_grand_product_device(g_ctx.poly.p, g_ctx.poly2.p, n, g_ctx.poly.p, s)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(z, g_ctx.poly.p, n * 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return ZKHIP_OK;
}

int zkhip_profile_enable(int on) {
  guard_t g(g_mu);
  g_prof_on = on != 0;
  g_prof_n = 0;
  return ZKHIP_OK;
}

int zkhip_profile_read(double* ms, char (*names)[64], int max) {
  guard_t g(g_mu);
  if (!g_prof_ev_ready || g_prof_n == 0) return 0;
  if (hipEventSynchronize(g_prof_ev[g_prof_n]) != hipSuccess) { set_error("profile_read: event sync failed"); return ZKHIP_EHIP; }
  for (int i = 0; i < g_prof_n && i < max; i++) {
    float t = 0;
    (void)hipEventElapsedTime(&t, g_prof_ev[i], g_prof_ev[i + 1]);
    if (ms) ms[i] = t;
    if (names) snprintf(names[i], 64, "%s", g_prof_names[i]);
  }
  return g_prof_n;
}

int zkhip_g1_fixed_base_mul_device(const void* d_scalars, size_t n, void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (n && (!d_scalars || !d_out)) { set_error("fixed_base_mul: null pointer"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = caller_stream(stream);
  if ((rc = g_ctx.ws.reserve(g1_fixed_base_workspace(n))) != ZKHIP_OK) return rc;
  if (!g_ctx.fixed_table_ready) {              // one-time: 16 windows x 2^15 multiples of the generator (32 MiB)
    if ((rc = g_ctx.fixed_table.reserve(g1_fixed_base_table_bytes())) != ZKHIP_OK) return rc;
    if ((rc = g1_fixed_base_table_build((uint32_t*)g_ctx.fixed_table.p, g_ctx.ws.p, g_ctx.ws.cap, s)) != ZKHIP_OK) return rc;
    g_ctx.fixed_table_ready = true;
  }
  return g1_fixed_base_mul_device((const uint32_t*)d_scalars, n, (const uint32_t*)g_ctx.fixed_table.p, (uint32_t*)d_out, g_ctx.ws.p, g_ctx.ws.cap, s);
}

int zkhip_g1_fft_device(void* d_points_xyz, const uint64_t omega[4], uint32_t log_n, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_points_xyz || !omega) { set_error("g1_fft: null pointer"); return ZKHIP_EINVAL; }
  if (log_n > 26) { set_error("g1_fft: log_n = %u out of range", log_n); return ZKHIP_EINVAL; }
  const size_t n = (size_t)1 << log_n;
  if ((rc = g_ctx.ws.reserve(g1_fft_workspace(n))) != ZKHIP_OK) return rc;
  return g1_fft_device((const uint32_t*)d_points_xyz, 1, (uint32_t*)d_points_xyz, 1, log_n, (const uint32_t*)omega, nullptr, g_ctx.ws.p, g_ctx.ws.cap,
                       caller_stream(stream));
}

int zkhip_g_to_lagrange_device(const void* d_g, uint32_t k, void* d_g_lagrange, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!d_g || !d_g_lagrange) { set_error("g_to_lagrange: null pointer"); return ZKHIP_EINVAL; }
  if (d_g == d_g_lagrange) { set_error("g_to_lagrange: the arrays may not alias"); return ZKHIP_EINVAL; }
  if (k > 26) { set_error("g_to_lagrange: k = %u out of range", k); return ZKHIP_EINVAL; }
  const size_t n = (size_t)1 << k;
  if ((rc = g_ctx.ws.reserve(g1_fft_workspace(n))) != ZKHIP_OK) return rc;
  namespace H = zkhip::halo2;
  H::Fr omega = H::fr_root_of_unity();
  for (uint32_t i = k; i < 28; i++) omega = H::detail::mul(omega, omega);
  const H::Fr omega_inv = H::detail::invert(omega), n_inv = H::detail::invert(H::detail::from_u64((uint64_t)n));
  return g1_fft_device((const uint32_t*)d_g, 0, (uint32_t*)d_g_lagrange, 0, k, (const uint32_t*)omega_inv.l, (const uint32_t*)n_inv.l, g_ctx.ws.p,
                       g_ctx.ws.cap, caller_stream(stream));
}

int zkhip_g1_gen_walk_device(const uint64_t t0[4], const uint64_t d[4], size_t n, void* d_out, void* stream) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (!t0 || !d || (n && !d_out)) { set_error("gen_walk: null pointer"); return ZKHIP_EINVAL; }
  if ((rc = g_ctx.ws.reserve(g1_gen_walk_workspace(n))) != ZKHIP_OK) return rc;
  return g1_gen_walk_device((const uint32_t*)t0, (const uint32_t*)d, n, (uint32_t*)d_out, g_ctx.ws.p, g_ctx.ws.cap,
                            caller_stream(stream));
}

// ---- parity hooks ----------------------------------------------------------------------------------
int zkhip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (field < 0 || field > 1 || op < 0 || op > 3 || (n && (!a || !b || !out))) { set_error("test_field_op: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = g_ctx.stream;
  dev_buf tmp;
  if ((rc = tmp.reserve(n * 96)) != ZKHIP_OK) return rc;
  char* d = (char*)tmp.p;
  hipError_t e = hipMemcpyAsync(d, a, n * 32, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d + n * 32, b, n * 32, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) rc = test_field_op(field, op, (uint32_t*)d, (uint32_t*)(d + n * 32), (uint32_t*)(d + n * 64), n, s);
  if (e == hipSuccess && rc == ZKHIP_OK) e = hipMemcpyAsync(out, d + n * 64, n * 32, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  tmp.release();
  if (e != hipSuccess) { set_error("test_field_op: %s", hipGetErrorString(e)); return ZKHIP_EHIP; }
  return rc;
}

int zkhip_test_g1_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out_xyz, size_t n) {
  guard_t g(g_mu);
  int rc = ensure_init();
  if (rc != ZKHIP_OK) return rc;
  if (op < 0 || op > 4 || (n && (!a || !b || !out_xyz))) { set_error("test_g1_op: bad argument"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  hipStream_t s = g_ctx.stream;
  dev_buf tmp;
  if ((rc = tmp.reserve(n * (64 + 64 + 96))) != ZKHIP_OK) return rc;
  char* d = (char*)tmp.p;
  hipError_t e = hipMemcpyAsync(d, a, n * 64, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d + n * 64, b, n * 64, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) rc = test_g1_op(op, (uint32_t*)d, (uint32_t*)(d + n * 64), (uint32_t*)(d + n * 128), n, s);
  if (e == hipSuccess && rc == ZKHIP_OK) e = hipMemcpyAsync(out_xyz, d + n * 128, n * 96, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  tmp.release();
  if (e != hipSuccess) { set_error("test_g1_op: %s", hipGetErrorString(e)); return ZKHIP_EHIP; }
  return rc;
}

}  // extern "C"
