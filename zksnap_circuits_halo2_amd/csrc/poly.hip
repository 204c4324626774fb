// Fr-vector primitives of the prover besides the NTT: `eval_polynomial`, `kate_division`
// ([DEP] halo2-axiom halo2_proofs/src/arithmetic.rs), `BatchInvert::batch_invert` ([DEP] ff crate, used by the
// permutation / lookup grand products) and the grand-product running product itself ([DEP] plonk/permutation/prover.rs:
// z[i+1] = z[i] * v[i]); all reached from create_proof, /root/reference/aggregator/src/wrapper.rs:129.
//
// All four are chunked linear recurrences: a thread owns CH consecutive elements, a first pass produces one aggregate
// per chunk, the aggregates are processed recursively (1/CH of the work), a second pass re-runs each chunk with its
// incoming carry.  2 multiplies per element, O(log_CH n) launches.
//   suffix Horner scan  out[i] = a[i] + b * out[i+1]           (kate_division; eval_polynomial is out[0])
//   prefix product      out[0] = 1, out[i+1] = out[i] * v[i]
//   batch inversion     Montgomery's trick per chunk with one inversion (division steps, fe_inverse.hpp) per thread
// Values are kept in the external Montgomery-256 domain throughout: x*2^256 is the radix-2^261 form of x*2^-5 and the
// recurrences are linear in the data, so only the constant multiplier is converted (fp29.hpp).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "fp29.hpp"
#include "fe_inverse.hpp"
#include "fr_vec.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

// elements per thread of the scans.  16, not 32 (round 3, same-box A/B in profiles/r03_batch_invert_ab.txt): 2^15 eval / kate / prefix 0.099 / 0.170 /
// 0.111 -> 0.086 / 0.139 / 0.078 ms, 2^20 0.130 / 0.231 / 0.160 -> 0.117 / 0.196 / 0.120, 2^24 0.367 / 0.980 / 0.854 -> 0.361 / 0.942 / 0.805 (8: the large
// sizes lose).  The batch inversion picks its own chunk length (INV_CH_MAX).
constexpr uint32_t POLY_CH = 16;
constexpr uint32_t INV_CH_MAX = 32;

// ---- suffix Horner scan ---------------------------------------------------------------------------
// pass A: agg[t] = sum_{i in chunk t} a[i] b^(i - lo)          (the chunk's scan value at its first element, carry-in 0)
__global__ void __launch_bounds__(256) k_horner_agg(const uint32_t* __restrict__ a, size_t n, const fe_arg* __restrict__ b_dev, uint32_t* __restrict__ agg) {
  const fe_arg b_ext = *b_dev;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * POLY_CH;
  if (lo >= n) return;
  const size_t hi = lo + POLY_CH < n ? lo + POLY_CH : n;
  const fe b = fr_const_internal(b_ext);
  fe q = fe_zero();
  size_t i = hi;
  // full blocks of four elements = one 128-byte line per lane: the eight loads of a block are issued together and the line is consumed at
  // once (element by element the same line was touched in four separate iterations, 64 lanes x 64 lines apart, and had to survive in L1)
  for (; i >= lo + 4 && ((i & 3) == 0); i -= 4) {
    uint32_t w0[8], w1[8], w2[8], w3[8];
    load_words(a + (i - 1) * 8, w3); load_words(a + (i - 2) * 8, w2); load_words(a + (i - 3) * 8, w1); load_words(a + (i - 4) * 8, w0);
    q = fe_norm(fe_add(fe_unpack<0>(w3), fe_mul<Fr>(b, q)));
    q = fe_norm(fe_add(fe_unpack<0>(w2), fe_mul<Fr>(b, q)));
    q = fe_norm(fe_add(fe_unpack<0>(w1), fe_mul<Fr>(b, q)));
    q = fe_norm(fe_add(fe_unpack<0>(w0), fe_mul<Fr>(b, q)));
  }
  for (; i-- > lo;) q = fe_norm(fe_add(load_ext(a, i), fe_mul<Fr>(b, q)));   // < 3p, N
  store_canon(agg, t, q);
}

// pass B: out[i] = a[i] + b * out[i+1] with carry-in carry[t+1] (the full scan value at the next chunk's first element)
__global__ void __launch_bounds__(256) k_horner_apply(const uint32_t* __restrict__ a, size_t n, const fe_arg* __restrict__ b_dev,
                                                      const uint32_t* __restrict__ carry, size_t ncarry, uint32_t* __restrict__ out,
                                                      size_t out_shift) {
  const fe_arg b_ext = *b_dev;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * POLY_CH;
  if (lo >= n) return;
  const size_t hi = lo + POLY_CH < n ? lo + POLY_CH : n;
  const fe b = fr_const_internal(b_ext);
  fe q = (carry != nullptr && t + 1 < ncarry) ? load_ext(carry, t + 1) : fe_zero();
  size_t i = hi;
  for (; i >= lo + 4 && ((i & 3) == 0); i -= 4) {               // four elements = one 128-byte line per lane, loaded together (see k_horner_agg)
    uint32_t w[4][8];
#pragma unroll
    for (int e = 0; e < 4; e++) load_words(a + (i - 4 + e) * 8, w[e]);
#pragma unroll
    for (int e = 3; e >= 0; e--) {
      q = fe_norm(fe_add(fe_unpack<0>(w[e]), fe_mul<Fr>(b, q)));
      if (i - 4 + e >= out_shift) store_canon(out, i - 4 + e - out_shift, q);     // kate_division drops the scan value at index 0
    }
  }
  for (; i-- > lo;) {
    q = fe_norm(fe_add(load_ext(a, i), fe_mul<Fr>(b, q)));
    if (i >= out_shift) store_canon(out, i - out_shift, q);
  }
}

// batched form of pass A for many polynomials of one length evaluated at one point (multiopen: every advice / fixed / permutation
// polynomial at x): blockIdx.y = polynomial; level 0 reads through a pointer table, deeper levels a dense [count][n] array.
// The last level (n <= CH) is the evaluation itself.
__global__ void __launch_bounds__(256) k_horner_agg_batch(const uint32_t* const* __restrict__ ptrs, const uint32_t* __restrict__ base, size_t n,
                                                          const fe_arg* __restrict__ b_dev, uint32_t* __restrict__ agg, size_t m) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * POLY_CH;
  if (lo >= n) return;
  const size_t hi = lo + POLY_CH < n ? lo + POLY_CH : n;
  const uint32_t* a = ptrs ? ptrs[blockIdx.y] : base + (size_t)blockIdx.y * n * 8;
  const fe_arg b_ext = *b_dev;
  const fe b = fr_const_internal(b_ext);
  fe q = fe_zero();
  size_t i = hi;
  for (; i >= lo + 4 && ((i & 3) == 0); i -= 4) {               // four elements = one 128-byte line per lane, loaded together (see k_horner_agg)
    uint32_t w[4][8];
#pragma unroll
    for (int e = 0; e < 4; e++) load_words(a + (i - 4 + e) * 8, w[e]);
#pragma unroll
    for (int e = 3; e >= 0; e--) q = fe_norm(fe_add(fe_unpack<0>(w[e]), fe_mul<Fr>(b, q)));
  }
  for (; i-- > lo;) q = fe_norm(fe_add(load_ext(a, i), fe_mul<Fr>(b, q)));
  store_canon(agg + (size_t)blockIdx.y * m * 8, t, q);
}

// ---- prefix product --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_prod_agg(const uint32_t* __restrict__ v, size_t n, uint32_t* __restrict__ agg) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * POLY_CH;
  if (lo >= n) return;
  const size_t hi = lo + POLY_CH < n ? lo + POLY_CH : n;
  uint32_t w[8];
  load_words(v + lo * 8, w);
  fe acc = fe_unpack<0>(w);                                   // x * 2^256 (external domain)
  size_t i = lo + 1;
  for (; i < hi && (i & 3) != 0; i++) {                       // up to the next 128-byte line
    load_words(v + i * 8, w);
    acc = fe_mul<Fr>(acc, fe_from_ext_lazy(w));               // ext-domain value times internal-form factor stays in the ext domain
  }
  for (; i + 4 <= hi; i += 4) {                               // four elements = one line per lane, loaded together (see k_horner_agg)
    uint32_t ww[4][8];
#pragma unroll
    for (int e = 0; e < 4; e++) load_words(v + (i + e) * 8, ww[e]);
#pragma unroll
    for (int e = 0; e < 4; e++) acc = fe_mul<Fr>(acc, fe_from_ext_lazy(ww[e]));
  }
  for (; i < hi; i++) {
    load_words(v + i * 8, w);
    acc = fe_mul<Fr>(acc, fe_from_ext_lazy(w));
  }
  store_canon(agg, t, acc);
}

// out[i] = carry[t] * prod_{lo <= j < i} v[j]   (exclusive); carry == nullptr: carry is 1
__global__ void __launch_bounds__(256) k_prod_apply(const uint32_t* __restrict__ v, size_t n, const uint32_t* __restrict__ carry,
                                                    uint32_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lo = t * POLY_CH;
  if (lo >= n) return;
  const size_t hi = lo + POLY_CH < n ? lo + POLY_CH : n;
  fe cur;
  if (carry) cur = load_ext(carry, t);
  else {                                                      // 1 in the external domain = 2^256 mod r
#pragma unroll
    for (int i = 0; i < NL; i++) cur.l[i] = Fr::TO_EXT[i];
  }
  uint32_t w[8];
  size_t i = lo;
  for (; i + 4 <= hi && (i & 3) == 0; i += 4) {               // four elements = one line per lane, loaded together (see k_horner_agg); all four
    uint32_t ww[4][8];                                        // are read before any of them is stored: v and out may alias
#pragma unroll
    for (int e = 0; e < 4; e++) load_words(v + (i + e) * 8, ww[e]);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      store_canon(out, i + e, cur);
      cur = fe_mul<Fr>(cur, fe_from_ext_lazy(ww[e]));
    }
  }
  for (; i < hi; i++) {
    load_words(v + i * 8, w);                                 // read before the store: v and out may alias
    const fe f = fe_from_ext_lazy(w);
    store_canon(out, i, cur);
    cur = fe_mul<Fr>(cur, f);
  }
}

// ---- batch inversion (zeros stay zero, like ff::BatchInvert) ------------------------------------------
// a^-1 in Fr, a internal & reduced: division steps instead of the Fermat chain a^(r-2) (fe_inverse.hpp)
__device__ __forceinline__ fe fr_inverse(const fe& a) { return fe_inverse<Fr>(a); }

// One wavefront per workgroup owns a tile of 64 * ch consecutive elements; lane l inverts the elements l, l + 64, l + 128, ... of the tile
// (Montgomery's trick groups ANY elements: with the lanes interleaved every load and store of the wavefront is one contiguous 2 KiB run,
// where consecutive elements per lane -- the layout of rounds 1-4 -- made each of them 64 separate 32-byte pieces 32 * ch bytes apart).
// The parked prefix products are limb-major (scratch[limb][element]) for the same reason.
__global__ void __launch_bounds__(64) k_batch_invert(uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ scratch, uint32_t ch) {
  const size_t first = (size_t)blockIdx.x * 64 * ch + threadIdx.x;
  if (first >= n) return;
  // x*2^256 read as the internal form of x' = x*2^-5.  prefix products of the non-zero x' (internal form), parked in scratch
  const fe one = fe_one<Fr>();
  fe pref = one;
  uint32_t w[8];
  uint32_t cnt = 0;
  for (size_t i = first; cnt < ch && i < n; i += 64, cnt++) {
    load_words(a + i * 8, w);
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) o |= w[k];
#pragma unroll
    for (int k = 0; k < NL; k++) scratch[(size_t)k * n + i] = pref.l[k];
    if (o) pref = fe_mul<Fr>(pref, fe_unpack<0>(w));
  }
  // inverse of the product, rescaled once so that the outputs come out in the external domain:
  // x'^-1 (internal) = x^-1 * 2^266; times 2^251 (Montgomery) -> x^-1 * 2^256
  fe c251 = fe_zero();
  c251.l[8] = 1u << (251 - 232);
  fe inv = fe_mul<Fr>(fr_inverse(pref), c251);
  for (uint32_t j = cnt; j-- > 0;) {
    const size_t i = first + (size_t)j * 64;
    load_words(a + i * 8, w);
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) o |= w[k];
    if (!o) continue;
    fe pre;
#pragma unroll
    for (int k = 0; k < NL; k++) pre.l[k] = scratch[(size_t)k * n + i];
    fe out = fe_mul<Fr>(inv, pre);
    inv = fe_mul<Fr>(inv, fe_unpack<0>(w));
    uint32_t wo[8];
    fe_pack(fe_canon_lt2p<Fr>(out), wo);
    store_words(a + i * 8, wo);
  }
}

// ---- host orchestration ------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

static inline size_t chunks_of(size_t n) { return (n + POLY_CH - 1) / POLY_CH; }
static inline dim3 grid_for(size_t threads, int block) { return dim3((unsigned)((threads + block - 1) / block)); }

size_t poly_workspace_bytes(size_t n) {   // aggregate arrays of every recursion level + constants + batch-invert scratch
  size_t total = 8192, m = n;
  while (m > 1) { m = chunks_of(m); total += ((m * 32 + 255) / 256) * 256 + 256; }
  return total + ((n * 36 + 255) / 256) * 256;
}

// out[i - out_shift] = sum_{j >= i} a[j] b^(j-i) for i >= out_shift (out may be nullptr: only the value at index 0 is
// wanted and is written to d_result).  b: the multiplier in device memory (8 external words).  ws: scratch.
static int horner_scan(const uint32_t* d_a, size_t n, const fe_arg* b, uint32_t* d_out, size_t out_shift,
                       uint32_t* d_result, char* ws, hipStream_t stream) {
  if (n <= POLY_CH) {   // one chunk
    if (d_out) hipLaunchKernelGGL(k_horner_apply, dim3(1), dim3(256), 0, stream, d_a, n, b, (const uint32_t*)nullptr, (size_t)0, d_out, out_shift);
    if (d_result) hipLaunchKernelGGL(k_horner_agg, dim3(1), dim3(256), 0, stream, d_a, n, b, d_result);
    HIPCHK(hipGetLastError());
    return ZKHIP_OK;
  }
  const size_t m = chunks_of(n);
  uint32_t* agg = (uint32_t*)ws;
  char* next_ws = ws + ((m * 32 + 255) / 256) * 256;
  hipLaunchKernelGGL(k_horner_agg, grid_for(m, 256), dim3(256), 0, stream, d_a, n, b, agg);
  // recursion: scan the aggregates in place with multiplier b^CH -- the next entry of the power array the caller parked (stage_const: b,
  // b^CH, b^(CH^2), ... computed by ONE thread up front; a launch per level for one power each cost 11 us apiece in the dependent chain);
  // the scan's value at index 0 is the overall result
  int rc = horner_scan(agg, m, b + 1, d_out ? agg : nullptr, 0, d_result, next_ws, stream);
  if (rc != ZKHIP_OK) return rc;
  if (d_out) hipLaunchKernelGGL(k_horner_apply, grid_for(m, 256), dim3(256), 0, stream, d_a, n, b, (const uint32_t*)agg, m, d_out, out_shift);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// a constant that arrives from the host travels as a kernel argument (by value: the caller's memory is free when the launch returns, and
// nothing waits for the stream -- a copy from the caller's memory would need a synchronisation) and is parked in device memory by one thread
// out[0] = v, out[l] = out[l - 1]^CH for l <= levels: the multipliers of every recursion level of a scan
constexpr uint32_t POW_LEVELS = 7;        // 16^7 = 2^28 elements; 8 constants = the 256 bytes the scans reserve
__global__ void k_store_const(fe_arg v, fe_arg* __restrict__ out, uint32_t levels) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = v;
  if (levels == 0) return;
  fe r = fr_const_internal(v);
  fe k;
#pragma unroll
  for (int i = 0; i < NL; i++) k.l[i] = Fr::TO_EXT[i];
#pragma unroll 1
  for (uint32_t l = 1; l <= levels; l++) {
    static_assert(POLY_CH == 16, "b^CH below is four squarings");
#pragma unroll
    for (int q = 0; q < 4; q++) r = fe_sqr<Fr>(r);              // b^16 (products of values < 2p stay < 2p)
    uint32_t w[8];
    fe_pack(fe_canon_lt2p<Fr>(fe_mul<Fr>(k, r)), w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[l].w[i] = w[i];
  }
}
static int store_const(const uint32_t host[8], fe_arg* d, hipStream_t stream, uint32_t levels = 0) {
  fe_arg v;
  memcpy(v.w, host, 32);
  hipLaunchKernelGGL(k_store_const, dim3(1), dim3(64), 0, stream, v, d, levels);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}
// recursion levels below the top one of a scan over n elements (= how many powers of the multiplier it needs)
static uint32_t scan_levels(size_t n) {
  uint32_t levels = 0;
  while (n > POLY_CH) { n = (n + POLY_CH - 1) / POLY_CH; levels++; }
  return levels < POW_LEVELS ? levels : POW_LEVELS;
}
// the multiplier of a scan over n elements and its powers: parked in the last 256 bytes of the workspace
static int stage_const(const uint32_t host[8], size_t n, char* ws, size_t ws_bytes, hipStream_t stream, fe_arg** out) {
  fe_arg* d = (fe_arg*)(ws + ws_bytes - 256);
  *out = d;
  return store_const(host, d, stream, scan_levels(n));
}

// eval_polynomial: result (8 words, device) = sum a[i] x^i
int fr_eval_polynomial_device(const uint32_t* d_a, size_t n, const uint32_t x_host[8], uint32_t* d_result, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  if (n == 0) { HIPCHK(hipMemsetAsync(d_result, 0, 32, stream)); return ZKHIP_OK; }
  if (ws_bytes < poly_workspace_bytes(n)) { set_error("eval_polynomial: workspace too small"); return ZKHIP_EINVAL; }
  fe_arg* b = nullptr;
  int rc = stage_const(x_host, n, (char*)ws, ws_bytes, stream, &b);
  if (rc != ZKHIP_OK) return rc;
  return horner_scan(d_a, n, b, nullptr, 0, d_result, (char*)ws, stream);
}

// `count` polynomials of n coefficients each, all evaluated at x: d_results[p] = sum_i poly_p[i] x^i.  One launch per recursion
// level for the whole batch (ceil(log_32 n) levels) instead of one recursion per polynomial.
size_t poly_batch_workspace_bytes(size_t n, size_t count) {
  size_t total = 8192 + ((count * 8 + 255) / 256) * 256, m = n;
  while (m > 1) { m = chunks_of(m); total += ((count * m * 32 + 255) / 256) * 256 + 256; }
  return total;
}

int fr_eval_polynomial_batch_device(const void* const* d_polys_host, size_t count, size_t n, const uint32_t x_host[8], uint32_t* d_results,
                                    void* ws, size_t ws_bytes, hipStream_t stream, arg_ring* ring) {
  if (count == 0) return ZKHIP_OK;
  if (n == 0) { HIPCHK(hipMemsetAsync(d_results, 0, count * 32, stream)); return ZKHIP_OK; }
  if (count > 65535) { set_error("eval_polynomial_batch: more than 65535 polynomials"); return ZKHIP_EINVAL; }
  if (ws_bytes < poly_batch_workspace_bytes(n, count)) { set_error("eval_polynomial_batch: workspace too small"); return ZKHIP_EINVAL; }
  char* p = (char*)ws;
  const uint32_t** d_ptrs = (const uint32_t**)p;
  p += ((count * 8 + 255) / 256) * 256;
  fe_arg* b = (fe_arg*)p;
  p += 256;
  int rc = upload_args(ring, d_ptrs, d_polys_host, count * 8, stream);        // caller memory: through the pinned ring, no wait for the stream
  if (rc == ZKHIP_OK) rc = store_const(x_host, b, stream, scan_levels(n));   // b, b^CH, b^(CH^2), ...: the multiplier of every level, one thread, once
  if (rc != ZKHIP_OK) return rc;
  const uint32_t* cur = nullptr;                            // level 0 reads through d_ptrs
  size_t cur_n = n;
  while (true) {
    const size_t m = chunks_of(cur_n);
    uint32_t* agg = m == 1 ? d_results : (uint32_t*)p;      // the last level's single aggregate per polynomial is the evaluation
    hipLaunchKernelGGL(k_horner_agg_batch, dim3((unsigned)((m + 255) / 256), (unsigned)count), dim3(256), 0, stream,
                       cur ? (const uint32_t* const*)nullptr : (const uint32_t* const*)d_ptrs, cur, cur_n, (const fe_arg*)b, agg, m);
    if (m == 1) break;
    p += ((count * m * 32 + 255) / 256) * 256 + 256;
    b = b + 1;
    cur = agg;
    cur_n = m;
  }
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// kate_division: q[i] = a[i+1] + b q[i+1], i < n-1  (quotient of a(X) by (X - b), remainder dropped)
int fr_kate_division_device(const uint32_t* d_a, size_t n, const uint32_t b_host[8], uint32_t* d_q, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
  if (n < 2) return ZKHIP_OK;
  if (ws_bytes < poly_workspace_bytes(n)) { set_error("kate_division: workspace too small"); return ZKHIP_EINVAL; }
  fe_arg* b = nullptr;
  int rc = stage_const(b_host, n, (char*)ws, ws_bytes, stream, &b);
  if (rc != ZKHIP_OK) return rc;
  // the suffix Horner scan of a at index i+1 is q[i]: scan everything, drop index 0
  return horner_scan(d_a, n, b, d_q, 1, nullptr, (char*)ws, stream);
}

static int prefix_product_rec(const uint32_t* d_v, size_t n, const uint32_t* carry, uint32_t* d_out, char* ws, hipStream_t stream);

// exclusive prefix product: out[0] = 1, out[i] = v[0] ... v[i-1]   (out may alias v)
int fr_prefix_product_device(const uint32_t* d_v, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (ws_bytes < poly_workspace_bytes(n)) { set_error("prefix_product: workspace too small"); return ZKHIP_EINVAL; }
  return prefix_product_rec(d_v, n, nullptr, d_out, (char*)ws, stream);
}

static int prefix_product_rec(const uint32_t* d_v, size_t n, const uint32_t* carry_unused, uint32_t* d_out, char* ws, hipStream_t stream) {
  (void)carry_unused;
  if (n <= POLY_CH) {
    hipLaunchKernelGGL(k_prod_apply, dim3(1), dim3(256), 0, stream, d_v, n, (const uint32_t*)nullptr, d_out);
    HIPCHK(hipGetLastError());
    return ZKHIP_OK;
  }
  const size_t m = chunks_of(n);
  uint32_t* agg = (uint32_t*)ws;
  char* next_ws = ws + ((m * 32 + 255) / 256) * 256 + 256;
  hipLaunchKernelGGL(k_prod_agg, grid_for(m, 256), dim3(256), 0, stream, d_v, n, agg);
  int rc = prefix_product_rec(agg, m, nullptr, agg, next_ws, stream);   // exclusive prefix products of the chunk products, in place
  if (rc != ZKHIP_OK) return rc;
  hipLaunchKernelGGL(k_prod_apply, grid_for(m, 256), dim3(256), 0, stream, d_v, n, (const uint32_t*)agg, d_out);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// ---- permutation argument: every set's grand product in one call ----------------------------------------------------------------
// [DEP] halo2-axiom plonk/permutation/prover.rs `Argument::commit` (reached from create_proof, /root/reference/aggregator/src/wrapper.rs:129):
// the permutation's columns are cut into sets of `chunk` (= degree - 2) columns; per set and row
//     den = prod_j (v_j[i] + beta sigma_j[i] + gamma)        num = prod_j (v_j[i] + beta delta^c omega^i + gamma),  c = the column's index
//     z_s[0] = z_(s-1)[u] (1 for the first set),  z_s[i + 1] = z_s[i] num[i] / den[i]
// with u the last usable row.  Only the rows below u enter any set's z[u], so with the ratio of the rows >= u forced to 1 the chained
// products of ALL sets are ONE exclusive prefix product over the [sets][n] array: set s starts at the product of the earlier sets' usable
// rows, which is z_(s-1)[u].  Three kernels of (sets x n) threads + the scan, whatever the number of sets (the state-transition and voter
// shapes of the reference have hundreds of columns at k = 13 .. 15: one call per set was 5 launches of 2^13 threads each, 135 times).

// table[i] = scale * base^(i << shift) in the INTERNAL form (x 2^261), canonical words: fe_unpack<0> of an entry is a reduced operand
__global__ void __launch_bounds__(256) k_pow_table_internal(fe_arg base, fe_arg scale, int has_scale, uint32_t shift, uint32_t count, uint32_t* __restrict__ table) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe b = fr_const_internal(base);
  for (uint32_t s = 0; s < shift; s++) b = fe_sqr<Fr>(b);
  fe v = fr_pow_u32(b, i);                                                          // < 2p, N
  if (has_scale) v = fe_mul<Fr>(v, fr_const_internal(scale));
  store_canon(table, i, v);
}

// z[s][i] = prod_j (v_j[i] + beta sigma_j[i] + gamma) over the columns of set s, external domain
__global__ void __launch_bounds__(256) k_perm_den(const uint32_t* const* __restrict__ values, const uint32_t* const* __restrict__ sigmas, uint32_t nperm,
                                                  uint32_t chunk, size_t n, const uint32_t* __restrict__ ctab, fe_arg gamma, uint32_t* __restrict__ z) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = blockIdx.y, lo = s * chunk, cnt = nperm - lo < chunk ? nperm - lo : chunk;
  const fe b = load_ext(ctab, 0), g_int = load_ext(ctab, 1);                          // beta, gamma in the internal form (canonical: < p, N)
  // first column in the external domain (beta_internal x sigma_external is external), the others in the internal one: external x internal
  // products stay external, so the result needs no conversion.  Sums of three: limbs < 3 * 2^29.
  fe acc = fe_add(fe_add(load_ext(values[lo], i), fe_mul<Fr>(b, load_ext(sigmas[lo], i))), fe_unpack<0>(gamma.w));     // < 4p
  for (uint32_t j = 1; j < cnt; j++) {
    uint32_t wv[8], ws[8];
    load_words(values[lo + j] + i * 8, wv);
    load_words(sigmas[lo + j] + i * 8, ws);
    const fe t = fe_add(fe_add(fe_unpack<5>(wv), fe_mul<Fr>(b, fe_unpack<5>(ws))), g_int);                              // < 36p
    acc = fe_mul<Fr>(fe_norm(acc), t);                                                                                 // 4 * 36 < 169: < 2p, N
  }
  if (cnt == 1) acc = fe_mul<Fr>(fe_norm(acc), fe_one<Fr>());                                                         // reduce the lone sum
  store_canon(z + (size_t)s * n * 8, i, acc);
}

// z[s][i] <- z[s][i] * prod_j (v_j[i] + (beta delta^c) omega^i + gamma) for i < usable, 1 for the rows from `usable` on
__global__ void __launch_bounds__(256) k_perm_num_mul(const uint32_t* const* __restrict__ values, uint32_t nperm, uint32_t chunk, size_t n, size_t usable,
                                                      const uint32_t* __restrict__ ctab, const uint32_t* __restrict__ dtab, const uint32_t* __restrict__ pow_lo,
                                                      const uint32_t* __restrict__ pow_hi, uint32_t h, uint32_t* __restrict__ z) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = blockIdx.y, lo = s * chunk, cnt = nperm - lo < chunk ? nperm - lo : chunk;
  uint32_t* zs = z + (size_t)s * n * 8;
  if (i >= usable) {                                                                  // 1 in the external domain = 2^256 mod r
    fe one;
#pragma unroll
    for (int k = 0; k < NL; k++) one.l[k] = Fr::TO_EXT[k];
    store_canon(zs, i, one);
    return;
  }
  const fe g_int = load_ext(ctab, 1);
  const fe x = fe_mul<Fr>(load_ext(pow_lo, i & ((1u << h) - 1)), load_ext(pow_hi, i >> h));   // omega^i, internal, < 2p (table entries are internal forms)
  fe acc = fe_zero();
  for (uint32_t j = 0; j < cnt; j++) {
    uint32_t wv[8];
    load_words(values[lo + j] + i * 8, wv);
    const fe t = fe_add(fe_add(fe_unpack<5>(wv), fe_mul<Fr>(load_ext(dtab, lo + j), x)), g_int);   // internal, < 36p, limbs < 3 * 2^29
    acc = j == 0 ? t : fe_mul<Fr>(fe_norm(acc), t);                                              // 36 * 36 / 169 + 1 < 9p, then < 3p, < 2p: N
  }
  store_canon(zs, i, fe_mul<Fr>(load_ext(zs, i), fe_norm(acc)));                                 // external (1 / den) x internal: external, < 2p
}

size_t perm_workspace_bytes(uint32_t nperm, uint32_t chunk, uint32_t log_n) {
  const size_t n = (size_t)1 << log_n, nsets = chunk ? (nperm + chunk - 1) / chunk : 0;
  const uint32_t h = (log_n + 1) / 2;
  const size_t tables = (((size_t)nperm * 16 + 255) / 256) * 256 + (((size_t)nperm * 32 + 255) / 256) * 256 + (((size_t)1 << h) + (n >> h)) * 32 + 768;
  return tables + poly_workspace_bytes(nsets * n);
}

int fr_permutation_products_device(const void* const* d_values_host, const void* const* d_sigmas_host, uint32_t nperm, uint32_t chunk, uint32_t log_n,
                                   size_t usable, const uint32_t beta[8], const uint32_t gamma[8], const uint32_t delta[8], const uint32_t omega[8],
                                   uint32_t* d_z, void* ws, size_t ws_bytes, hipStream_t stream, arg_ring* ring) {
  if (nperm == 0) return ZKHIP_OK;
  const size_t n = (size_t)1 << log_n, nsets = (nperm + chunk - 1) / chunk;
  if (nsets > 65535) { set_error("permutation_products: more than 65535 sets"); return ZKHIP_EINVAL; }
  if (ws_bytes < perm_workspace_bytes(nperm, chunk, log_n)) { set_error("permutation_products: workspace too small"); return ZKHIP_EINVAL; }
  const uint32_t h = (log_n + 1) / 2;
  char* p = (char*)ws;
  const uint32_t** d_ptrs = (const uint32_t**)p;                  // [values ..., sigmas ...]
  p += (((size_t)nperm * 16 + 255) / 256) * 256;
  uint32_t* dtab = (uint32_t*)p;
  p += (((size_t)nperm * 32 + 255) / 256) * 256;
  uint32_t* pow_lo = (uint32_t*)p;
  p += ((size_t)1 << h) * 32;
  uint32_t* pow_hi = (uint32_t*)p;
  p += (n >> h) * 32;
  uint32_t* ctab = (uint32_t*)p;                                  // beta, gamma in the internal form: converted once, not per row
  p += 64;
  p = (char*)ws + ((((size_t)(p - (char*)ws)) + 255) / 256) * 256;
  const size_t rest = ws_bytes - (size_t)(p - (char*)ws);
  int rc = upload_args(ring, d_ptrs, d_values_host, (size_t)nperm * 8, stream);          // caller memory: through the pinned ring, no wait for the stream
  if (rc == ZKHIP_OK) rc = upload_args(ring, d_ptrs + nperm, d_sigmas_host, (size_t)nperm * 8, stream);
  if (rc != ZKHIP_OK) return rc;
  fe_arg b, g, d, w;
  memcpy(b.w, beta, 32); memcpy(g.w, gamma, 32); memcpy(d.w, delta, 32); memcpy(w.w, omega, 32);
  hipLaunchKernelGGL(k_pow_table_internal, dim3(1), dim3(256), 0, stream, d, b, 1, 0u, 1u, ctab);            // delta^0 beta
  hipLaunchKernelGGL(k_pow_table_internal, dim3(1), dim3(256), 0, stream, d, g, 1, 0u, 1u, ctab + 8);        // delta^0 gamma
  hipLaunchKernelGGL(k_pow_table_internal, grid_for(nperm, 256), dim3(256), 0, stream, d, b, 1, 0u, nperm, dtab);
  hipLaunchKernelGGL(k_pow_table_internal, grid_for((size_t)1 << h, 256), dim3(256), 0, stream, w, w, 0, 0u, 1u << h, pow_lo);
  hipLaunchKernelGGL(k_pow_table_internal, grid_for(n >> h, 256), dim3(256), 0, stream, w, w, 0, h, (uint32_t)(n >> h), pow_hi);
  const dim3 grid((unsigned)((n + 255) / 256), (unsigned)nsets);
  hipLaunchKernelGGL(k_perm_den, grid, dim3(256), 0, stream, (const uint32_t* const*)d_ptrs, (const uint32_t* const*)(d_ptrs + nperm), nperm, chunk, n, (const uint32_t*)ctab, g, d_z);
  HIPCHK(hipGetLastError());
  prof_mark(stream, "perm_den");
  rc = fr_batch_invert_device(d_z, nsets * n, p, rest, stream);
  if (rc != ZKHIP_OK) return rc;
  prof_mark(stream, "perm_invert");
  hipLaunchKernelGGL(k_perm_num_mul, grid, dim3(256), 0, stream, (const uint32_t* const*)d_ptrs, nperm, chunk, n, usable, (const uint32_t*)ctab, (const uint32_t*)dtab,
                     (const uint32_t*)pow_lo, (const uint32_t*)pow_hi, h, d_z);
  HIPCHK(hipGetLastError());
  prof_mark(stream, "perm_num");
  rc = fr_prefix_product_device(d_z, nsets * n, d_z, p, rest, stream);
  prof_mark(stream, "perm_scan");
  return rc;
}

// ---- linear combination of many columns: out[i] = sum_j c_j col_j[i] -------------------------------------------------------------------
// The y- / v-combinations and L(X) of the multi-open provers ([DEP] poly/kzg/multiopen/shplonk/prover.rs: hundreds of polynomials at the
// voter / state-transition column counts).  As a row program this is a chain of `count` dependent multiply-adds per row -- at 2^13 rows
// 128 wavefronts walking 650 instructions one after the other.  Here the columns are cut into groups of LC_GROUP: thread (row, group) adds
// its group's products two at a time under ONE Montgomery reduction (fe_mul_add), a second kernel adds the groups' partial sums; with one
// group the first kernel writes the result itself.  Coefficients are converted to the internal form once, by `count` threads.
constexpr uint32_t LC_GROUP = 32;

__global__ void __launch_bounds__(256) k_to_internal(const uint32_t* __restrict__ ext, uint32_t count, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe_arg c;
#pragma unroll
  for (int k = 0; k < 8; k++) c.w[k] = ext[(size_t)i * 8 + k];
  store_canon(out, i, fr_const_internal(c));
}

__global__ void __launch_bounds__(256) k_lincomb(const uint32_t* const* __restrict__ cols, const uint32_t* __restrict__ coeff_int, uint32_t count, size_t n,
                                                 uint32_t* __restrict__ out /* [groups][n] */) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t lo = blockIdx.y * LC_GROUP, hi = lo + LC_GROUP < count ? lo + LC_GROUP : count;
  fe acc = fe_zero();
  uint32_t j = lo;
  for (; j + 1 < hi; j += 2) {          // internal-form coefficient x external value = external; both operands N-form < p: each pair < 2p
    const fe t = fe_mul_add<Fr>(load_ext(coeff_int, j), load_ext(cols[j], i), load_ext(coeff_int, j + 1), load_ext(cols[j + 1], i));
    acc = fe_norm(fe_add(acc, t));
  }
  if (j < hi) acc = fe_norm(fe_add(acc, fe_mul<Fr>(load_ext(coeff_int, j), load_ext(cols[j], i))));
  store_canon(out + (size_t)blockIdx.y * n * 8, i, fe_reduce_soft<Fr>(acc));       // < 34p < 2^261 -> < 2p + 2^233 -> canonical
}

// out[i] = sum_g partial[g][i]
__global__ void __launch_bounds__(256) k_sum_columns(const uint32_t* __restrict__ partial, uint32_t groups, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe acc = load_ext(partial, i);
  for (uint32_t g = 1; g < groups; g++) {
    acc = fe_add(acc, load_ext(partial + (size_t)g * n * 8, i));
    if ((g & 3) == 3) acc = fe_norm(acc);           // four more canonical values per round: limbs stay below 2^32
    if ((g & 63) == 63) acc = fe_reduce_soft<Fr>(acc);     // ... and the value below 2^261 (~169 p) whatever the number of groups
  }
  store_canon(out, i, fe_reduce_soft<Fr>(fe_norm(acc)));
}

size_t lincomb_workspace_bytes(size_t count, size_t n) {
  const size_t groups = (count + LC_GROUP - 1) / LC_GROUP;
  return ((count * 8 + 255) / 256) * 256 + 2 * ((count * 32 + 255) / 256) * 256 + (groups > 1 ? groups * n * 32 : 0) + 256;
}

int fr_linear_combination_device(const void* const* d_cols_host, const uint32_t* coeffs_host, size_t count, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes,
                                 hipStream_t stream, arg_ring* ring) {
  if (n == 0) return ZKHIP_OK;
  if (count == 0) { HIPCHK(hipMemsetAsync(d_out, 0, n * 32, stream)); return ZKHIP_OK; }
  const size_t groups = (count + LC_GROUP - 1) / LC_GROUP;
  if (groups > 65535) { set_error("linear_combination: more than %u columns", 65535u * LC_GROUP); return ZKHIP_EINVAL; }
  if (ws_bytes < lincomb_workspace_bytes(count, n)) { set_error("linear_combination: workspace too small"); return ZKHIP_EINVAL; }
  char* p = (char*)ws;
  const uint32_t** d_ptrs = (const uint32_t**)p;
  p += ((count * 8 + 255) / 256) * 256;
  uint32_t* d_ext = (uint32_t*)p;
  p += ((count * 32 + 255) / 256) * 256;
  uint32_t* d_int = (uint32_t*)p;
  p += ((count * 32 + 255) / 256) * 256;
  uint32_t* partial = (uint32_t*)p;
  int rc = upload_args(ring, d_ptrs, d_cols_host, count * 8, stream);          // caller memory: through the pinned ring, no wait for the stream
  if (rc == ZKHIP_OK) rc = upload_args(ring, d_ext, coeffs_host, count * 32, stream);
  if (rc != ZKHIP_OK) return rc;
  hipLaunchKernelGGL(k_to_internal, grid_for(count, 256), dim3(256), 0, stream, (const uint32_t*)d_ext, (uint32_t)count, d_int);
  hipLaunchKernelGGL(k_lincomb, dim3((unsigned)((n + 255) / 256), (unsigned)groups), dim3(256), 0, stream, (const uint32_t* const*)d_ptrs, (const uint32_t*)d_int,
                     (uint32_t)count, n, groups > 1 ? partial : d_out);
  if (groups > 1) hipLaunchKernelGGL(k_sum_columns, grid_for(n, 256), dim3(256), 0, stream, (const uint32_t*)partial, (uint32_t)groups, n, d_out);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

int fr_batch_invert_device(uint32_t* d_a, size_t n, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (ws_bytes < poly_workspace_bytes(n)) { set_error("batch_invert: workspace too small"); return ZKHIP_EINVAL; }
  // elements per thread: 3 multiplications each + one inversion (~50 multiplications' worth since it is division steps, fe_inverse.hpp).
  // Full chunks when there are enough of them to fill the chip; shorter ones below that, where the call is the latency of one thread.
  uint32_t ch = INV_CH_MAX;
  while (ch > 4 && n / ch < 65536) ch >>= 1;
  // one wavefront per workgroup: with few waves per CU, 128-thread workgroups land pairwise on the same two SIMDs (measured: 1024 waves take
  // 0.098 ms as 512 workgroups of 128 threads, 0.063 ms as 1024 of 64 or 256 of 256 -- profiles/r03_batch_invert_ab.txt)
  hipLaunchKernelGGL(k_batch_invert, grid_for((n + ch - 1) / ch, 64), dim3(64), 0, stream, d_a, n, (uint32_t*)ws, ch);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

}  // namespace zkhip
