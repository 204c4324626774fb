// BN254 G2 multi-scalar multiplication: `best_multiexp::<G2Affine>` [DEP halo2-axiom arithmetic.rs is generic over the curve].
//
// The north star names "Pippenger-bucketed MSM on BN254 G1/G2"; the reference prover itself never multiplies in G2 (it reads
// `params.g2()` / `params.s_g2()`: /root/reference/aggregator/src/wrapper.rs:1142-1144), so this path is built for correctness and
// shares everything that depends on the scalars only with the G1 path (msm.hip `msm_build_tasks`: signed digits, LDS counting sorts,
// tasks in longest-first order).  Curve-specific, here: bucket accumulation (one thread per task, XYZZ over Fq2 in registers),
// per-bucket combination of the task partials, the log-depth bucket pyramid (same index algebra as k_pyramid_step in msm.hip), the
// per-window weighted sum and the window fold.  One bucket set per window (no prepared tables: there is no fixed-base caller).
// Work points are 288 bytes (8 x 9 limbs).
//
// Round 5: the general-path machinery of G1 applies unchanged, because it depends on the scalars only -- GLV digits (the twist E': y^2 = x^3 + 3 / (9 + u)
// has j-invariant 0 as well: phi(x, y) = (beta' x, y) is an endomorphism, and on the order-r subgroup phi = lambda for the SAME lambda as on G1 when
// beta' = beta^2, beta the constant of msm.hip glv::BETA_EXT -- checked against the big-integer oracle in tests/test_oracle.py), hence 2 n points,
// ceil(128 / c) windows of c = 16 bits at every size from 2^12 up, the two-level LDS sort from 2^18 points, and a window fold of 128 - c dependent
// doublings instead of 256 - c (240 quad doublings at ~8 us were 1.9 ms of a 5.8 ms MSM at 2^16).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "ec2.hpp"
#include "ec2_quad.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

struct task_t {
  uint32_t bucket, start, len;
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

// beta^2 (Montgomery-256 words): the cube root of unity in Fq with (beta^2 x, y) = lambda (x, y) on G2 for the lambda of msm.hip's GLV split
__device__ constexpr uint32_t BETA2_EXT[8] = {0x13e80b9cu, 0x3350c88eu, 0xdb5e56b9u, 0x7dce557cu, 0xb615564au, 0x6001b4b8u, 0x020217e0u, 0x2682e617u};

// endo[i] = phi(bases[i]) = (beta^2 x.c0, beta^2 x.c1, y) in the G2Affine memory format; the identity (all zero) stays the identity
__global__ void __launch_bounds__(256) k2_endo_bases(const uint32_t* __restrict__ bases, uint32_t n, uint32_t* __restrict__ endo) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine2_words pt = load_affine2(bases, i);
  const fe beta = fq_from_ext(BETA2_EXT);
  uint32_t out[32];
  fq_to_ext(fe_mul<Fq>(beta, fq_from_ext(pt.w)), out);
  fq_to_ext(fe_mul<Fq>(beta, fq_from_ext(pt.w + 8)), out + 8);
#pragma unroll
  for (int k = 16; k < 32; k++) out[k] = pt.w[k];
  uint4* o = reinterpret_cast<uint4*>(endo + (size_t)i * 32);
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}

// point references >= split (GLV digits) are the endomorphism images, a second array (bases2[ref - split]).
// Round 5: lazy Fq2 arithmetic (ec2.hpp xyzz2_madd_lazy: fused product pairs, bound-tracked additions) with single-chain product columns, the
// coordinates unpacked lazily (no multiplication on load) and the next point gathered under the current addition.
__global__ void __launch_bounds__(64) k2_accumulate(const uint32_t* __restrict__ ntasks_p, const uint4* __restrict__ order, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bases, uint32_t* __restrict__ partials, uint32_t* __restrict__ buckets,
                                                    const uint32_t* __restrict__ bases2, uint32_t split) {
  auto point = [&](uint32_t ref) -> affine2_words {
    const uint32_t idx = ref & 0x7fffffffu;
    return idx >= split ? load_affine2(bases2, idx - split) : load_affine2(bases, idx);
  };
  const uint32_t ntasks = *ntasks_p;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ntasks; i += gridDim.x * blockDim.x) {
    const uint4 rec = order[i];
    const uint32_t t = rec.x, start = rec.y, len = rec.z;
    xyzz2 acc = xyzz2_identity();
    uint32_t ref = rec.w;                                        // the task's first reference rides in the order record
    affine2_words pt = point(ref);
#pragma unroll 1
    for (uint32_t j = 0; j < len; j++) {
      const affine2_words cur = pt;
      const uint32_t cref = ref;
      if (j + 1 < len) { ref = sorted[start + j + 1]; pt = point(ref); }
      if (affine2_is_identity(cur)) continue;
      uint32_t w[8];
      fe2 x, y;
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = cur.w[k];
      x.c0 = fe_from_ext_lazy(w);
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = cur.w[8 + k];
      x.c1 = fe_from_ext_lazy(w);
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = cur.w[16 + k];
      y.c0 = fe_from_ext_lazy(w);
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = cur.w[24 + k];
      y.c1 = fe_from_ext_lazy(w);
      xyzz2_madd_lazy<true>(acc, x, y, (cref >> 31) != 0);
    }
    acc = xyzz2_to_std(acc);
    if (t >> 31) store_xyzz2(buckets, t & 0x7fffffffu, acc);      // the bucket's only task writes the bucket itself
    else store_xyzz2(partials, t, acc);
  }
}

__device__ __forceinline__ xyzz2 xyzz2_shfl_xor(const xyzz2& a, int mask) {
  xyzz2 r;
  const fe* src[8] = {&a.X.c0, &a.X.c1, &a.Y.c0, &a.Y.c1, &a.ZZ.c0, &a.ZZ.c1, &a.ZZZ.c0, &a.ZZZ.c1};
  fe* dst[8] = {&r.X.c0, &r.X.c1, &r.Y.c0, &r.Y.c1, &r.ZZ.c0, &r.ZZ.c1, &r.ZZZ.c0, &r.ZZZ.c1};
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int i = 0; i < NL; i++) dst[k]->l[i] = (uint32_t)__shfl_xor((int)src[k]->l[i], mask, 64);
  return r;
}

// LANES adjacent lanes per bucket (1, 2, 4 or 8: msm_lay_out picks it from the expected partials per bucket, as for G1): lane q sums the task
// partials q, q + LANES, ..., log2(LANES) xor-shuffle steps add the lane sums (a bucket with a single task was written by k2_accumulate).
// A short top window puts thousands of entries into a handful of buckets: with one thread per bucket their partials were a chain of 256
// dependent additions (5 ms of a 19 ms MSM at 2^16).  Round 5: the lane count follows the layout (always 8 before: a bucket of two partials
// then paid one addition and three shuffle-tree additions on eight lanes -- 2.9 ms of a 6.2 ms MSM at 2^18) and the additions are the lazy ones.
template <int LANES>
__global__ void __launch_bounds__(64) k2_combine(const uint32_t* __restrict__ task_off, uint32_t nbuckets, const uint32_t* __restrict__ partials,
                                                 uint32_t* __restrict__ buckets, uint32_t seq_parts) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = gid / LANES, q = gid % LANES;
  const bool live = k < nbuckets;                 // whole groups of LANES lanes are live or dead together
  xyzz2 acc = xyzz2_identity();
  uint32_t m = 0;
  if (live) {
    const uint32_t t = task_off[k];
    m = task_off[k + 1] - t;
    if (m > seq_parts) m = 1;                     // a heavy bucket: k2_combine_heavy's (treated here like the single-task case: nothing to do)
    if (m != 1) {
#pragma unroll 1
      for (uint32_t j = q; j < m; j += LANES) acc = xyzz2_add_lazy_regmem<true>(acc, partials, t + j);
    }
  }
#pragma unroll 1
  for (int mask = 1; mask < LANES; mask <<= 1) acc = xyzz2_add_lazy<true>(acc, xyzz2_shfl_xor(acc, mask));
  if (live && q == 0 && m != 1) store_xyzz2(buckets, k, acc);
}

// heavy buckets (more than seq_parts partials: the list k_make_tasks collects, msm.hip section 7; a short top window makes a few dozen of
// them): a wavefront per bucket -- 64 lanes sum the partials 64 apart, an LDS tree adds the lane sums.  The list holds one entry per
// 2048-partial slice; the slice-0 entry stands for the whole bucket here (G2 MSMs are small).
__global__ void __launch_bounds__(64) k2_combine_heavy(const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ partials, uint32_t* __restrict__ buckets,
                                                       const uint32_t* __restrict__ heavy_count, const uint2* __restrict__ heavy, uint32_t heavy_cap) {
  __shared__ __attribute__((aligned(16))) uint32_t xch[32 * 72];
  const uint32_t nh = min(*heavy_count, heavy_cap);
  for (uint32_t i = blockIdx.x; i < nh; i += gridDim.x) {          // uniform over the workgroup
    const uint2 e = heavy[i];
    if (e.y != 0) continue;
    const uint32_t k = e.x, t = task_off[k], m = task_off[k + 1] - t;
    xyzz2 acc = xyzz2_identity();
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < m; j += 64) acc = xyzz2_add_lazy_regmem<true>(acc, partials, t + j);
#pragma unroll 1
    for (uint32_t half = 32; half >= 1; half >>= 1) {
      __syncthreads();
      if (threadIdx.x >= half && threadIdx.x < 2 * half) store_xyzz2(xch, threadIdx.x - half, acc);
      __syncthreads();
      if (threadIdx.x < half) acc = xyzz2_add_lazy_regmem<true>(acc, xch, threadIdx.x);
    }
    if (threadIdx.x == 0) store_xyzz2(buckets, k, acc);
  }
}

// pyramid step (see msm.hip section 8 for the state layout): sum_k (k + 1) B_k = Tot + sum_l 2^l T_l
__global__ void __launch_bounds__(64) k2_pyramid_step(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t N, int s, uint32_t in_stride,
                                                      uint32_t out_stride) {
  const uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= per_win) return;
  const int win = blockIdx.y;
  const uint32_t* wi = in + (size_t)win * in_stride * 72;
  uint32_t* wo = out + (size_t)win * out_stride * 72;
  uint32_t ia, ib;
  if (tid < N / 2) {
    ia = 2 * tid; ib = 2 * tid + 1;
  } else {
    const uint32_t r = tid - N / 2, l = r / (N / 4), u = r % (N / 4);
    if ((int)l == s - 1) { ia = 4 * u + 1; ib = 4 * u + 3; }
    else { ia = N + l * (N / 2) + 2 * u; ib = ia + 1; }
  }
  store_xyzz2(wo, tid, xyzz2_add_lazy_mem<true>(wi, ia, wi, ib));
}

// the same step with one quad per addition, for steps far smaller than the chip (their time is the latency of one addition)
__global__ void __launch_bounds__(64) k2_pyramid_step_quad(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t N, int s, uint32_t in_stride,
                                                           uint32_t out_stride) {
  const uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tid = gid >> 2, q = gid & 3;
  if (tid >= per_win) return;                     // whole quads leave together
  const int win = blockIdx.y;
  const uint32_t* wi = in + (size_t)win * in_stride * 72;
  uint32_t* wo = out + (size_t)win * out_stride * 72;
  uint32_t ia, ib;
  if (tid < N / 2) {
    ia = 2 * tid; ib = 2 * tid + 1;
  } else {
    const uint32_t r = tid - N / 2, l = r / (N / 4), u = r % (N / 4);
    if ((int)l == s - 1) { ia = 4 * u + 1; ib = 4 * u + 3; }
    else { ia = N + l * (N / 2) + 2 * u; ib = ia + 1; }
  }
  const xyzz2 r = xyzz2_quad_to_std(xyzz2_add_quad_lazy(load_xyzz2(wi, ia), load_xyzz2(wi, ib), q));      // stored points are in standard form
  if (q == 0) store_xyzz2(wo, tid, r);
}

// window sum = X0 + X1 + sum_l 2^l Z^l + 2^nz X1 (state after the last pyramid step: X has 2 elements, Z^0 .. Z^(nz-1) one each).
// One 32-lane group per window, a shuffle tree over the terms (nz + 2 <= 32).
__global__ void __launch_bounds__(128) k2_window_sum(const uint32_t* __restrict__ in, uint32_t in_stride, int nz, uint32_t* __restrict__ winsum, int W) {
  // one quad per term (32 terms, two wavefronts) per window; doubling chains and the addition tree on the quad formulas; the two
  // wavefronts meet through LDS for the last addition
  __shared__ __attribute__((aligned(16))) uint32_t xch[72];
  const int win = blockIdx.x, term = threadIdx.x >> 2;
  const uint32_t q = threadIdx.x & 3;
  const uint32_t* wi = in + (size_t)win * in_stride * 72;
  xyzz2 acc = xyzz2_identity();
  int dbl = 0;
  if (term < nz) { acc = load_xyzz2(wi, 2 + term); dbl = term; }
  else if (term == nz) { acc = load_xyzz2(wi, 1); dbl = nz; }
  else if (term == nz + 1) acc = xyzz2_add_quad_lazy(load_xyzz2(wi, 0), load_xyzz2(wi, 1), q);
  // the chains run on the lazy quad formulas (ec2_quad.hpp): X, Y standard form, ZZ / ZZZ below 5p between the steps
#pragma unroll 1
  for (int i = 0; i < dbl; i++) acc = xyzz2_dbl_quad_lazy(acc, q);
#pragma unroll 1
  for (int mask = 4; mask < 64; mask <<= 1) acc = xyzz2_add_quad_lazy(acc, xyzz2_quad_to_std(xyzz2_shfl_xor(acc, mask)), q);
  acc = xyzz2_quad_to_std(acc);
  if (threadIdx.x == 64) store_xyzz2(xch, 0, acc);      // sum of terms 16..31
  __syncthreads();
  if (threadIdx.x < 4) {
    acc = xyzz2_quad_to_std(xyzz2_add_quad_lazy(acc, load_xyzz2(xch, 0), q));
    if (q == 0) store_xyzz2(winsum, win, acc);
  }
}

// result = sum_w 2^(c w) winsum[w], written as a Jacobian G2 point (48 words)
__global__ void __launch_bounds__(64) k2_fold(const uint32_t* __restrict__ winsum, int W, int c, uint32_t* __restrict__ out) {
  // one quad: c (W - 1) dependent doublings are the whole cost.  (As one thread this was uniform work, which the compiler moved to the
  // scalar unit -- no 32 x 32 -> 64 multiply-add there: 48 us per doubling, 12 ms of a 19 ms MSM; kept on the vector unit 20 us; on the
  // quad formulas ~8 us.)
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  const uint32_t q = threadIdx.x;
  xyzz2 acc = load_xyzz2(winsum, W - 1);
  for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int i = 0; i < c; i++) acc = xyzz2_dbl_quad_lazy(acc, q);
    acc = xyzz2_add_quad_lazy(acc, load_xyzz2(winsum, w), q);
  }
  if (q == 0) store_jacobian2(xyzz2_quad_to_std(acc), out);
}

__global__ void __launch_bounds__(64) k2_store_identity(uint32_t* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) store_jacobian2(xyzz2_identity(), out);
}

// parity hook for the quad formulas: 2 a[i] + b[i] (op 3), 4 a[i] (op 4), and the lazy chains 4 a[i] + b[i] (op 5), 16 a[i] (op 6); four lanes per row
__global__ void __launch_bounds__(64) k2_test_op_quad(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, size_t n) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gid >> 2;
  const uint32_t q = (uint32_t)(gid & 3);
  if (i >= n) return;
  const affine2_words pa = load_affine2(a, i), pb = load_affine2(b, i);
  xyzz2 A = xyzz2_identity(), B = xyzz2_identity();
  fe2 x, y;
  if (!affine2_is_identity(pa)) { affine2_coords(pa, false, x, y); xyzz2_madd(A, x, y); A = xyzz2_dbl(A); }     // (a general representative: 2 a[i])
  if (!affine2_is_identity(pb)) { affine2_coords(pb, false, x, y); xyzz2_madd(B, x, y); }
  xyzz2 r;
  if (op == 3) r = xyzz2_add_quad(A, B, q);                                         // 2 a[i] + b[i]
  else if (op == 4) r = xyzz2_dbl_quad(A, q);                                       // 4 a[i]
  else if (op == 5) r = xyzz2_quad_to_std(xyzz2_add_quad_lazy(xyzz2_dbl_quad_lazy(A, q), B, q));                       // lazy chain: 4 a[i] + b[i]
  else r = xyzz2_quad_to_std(xyzz2_dbl_quad_lazy(xyzz2_dbl_quad_lazy(xyzz2_dbl_quad_lazy(A, q), q), q));                // lazy chain: 16 a[i]
  if (q == 0) store_jacobian2(r, out + i * 48);
}

// parity hook: out[i] = a[i] + b[i] (op 0), 2 a[i] (op 1), a[i] - b[i] (op 2) as Jacobian G2 points
__global__ void __launch_bounds__(64) k2_test_op(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine2_words pa = load_affine2(a, i), pb = load_affine2(b, i);
  xyzz2 acc = xyzz2_identity();
  fe2 x, y;
  if (!affine2_is_identity(pa)) { affine2_coords(pa, false, x, y); xyzz2_madd(acc, x, y); }
  if (op == 1) acc = xyzz2_dbl(acc);
  else if (!affine2_is_identity(pb)) {
    affine2_coords(pb, op == 2, x, y);
    // through the general addition as well as the mixed one: even rows use xyzz2_add on a lifted point
    if (i & 1) xyzz2_madd(acc, x, y);
    else { xyzz2 t = xyzz2_identity(); xyzz2_madd(t, x, y); acc = xyzz2_add(acc, t); }
  }
  store_jacobian2(acc, out + i * 48);
}

static int pick_window_g2(size_t n, bool glv) {
  // the model of msm_pick_window (W windows x (n' mixed additions + 2 general additions per bucket)) with G2's costs in G1 multiplications:
  // a mixed addition 8 M2 + 2 S2 ~ 28 + the Fq2 additions ~ 34, a general one ~ 48; plus the fold's c (W - 1) dependent quad doublings at
  // ~8 us each, i.e. ~1.3 M multiplication slots of the accumulation kernel per doubling (what makes small windows lose at small n)
  int best = 2;
  double best_cost = 1e300;
  for (int c = 2; c <= 16; c++) {
    const int W = glv ? msm_glv_windows(c) : (256 + c - 1) / c, top_bits = (glv ? 127 : 254) - (W - 1) * c;
    if (top_bits > 0 && top_bits < 6) continue;
    const double cost = (double)W * (34.0 * (double)(glv ? 2 * n : n) + 96.0 * (double)(1u << (c - 1))) + 1.3e6 * (double)c * (double)(W - 1) / 16.0;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

size_t msm_g2_workspace_bytes(size_t n) { return n ? msm_workspace_bytes(n, pick_window_g2(n, msm_uses_glv(false, 1, 288)), false, 1, 288) : 0; }

// d_scalars: n x 8 words (Fr, Montgomery), d_bases: n x 32 words (G2Affine), d_out: 48 words (G2 Jacobian)
int msm_g2_device(const uint32_t* d_scalars, const uint32_t* d_bases, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) {
    hipLaunchKernelGGL(k2_store_identity, dim3(1), dim3(64), 0, stream, d_out);
    HIPCHK(hipGetLastError());
    return ZKHIP_OK;
  }
  if (n >= (1ull << 31)) { set_error("msm_g2: n = %zu too large", n); return ZKHIP_EINVAL; }
  const bool glv = msm_uses_glv(false, 1, 288);
  const int c = pick_window_g2(n, glv);
  msm_tasks_view tv;
  int rc = msm_build_tasks(d_scalars, n, 1, n, c, false, 0u, 0u, 288, ws, ws_bytes, stream, &tv, glv);
  if (rc != ZKHIP_OK) return rc;
  if (glv) hipLaunchKernelGGL(k2_endo_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_bases, (uint32_t)n, tv.endo);
  {
    uint32_t blocks = (uint32_t)((tv.max_tasks + 63) / 64);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k2_accumulate, dim3(blocks), dim3(64), 0, stream, tv.ntasks, tv.order, tv.sorted, d_bases, tv.partials, tv.pyrA, (const uint32_t*)tv.endo,
                       glv ? (uint32_t)n : 0xffffffffu);
  }
  prof_mark(stream, "accumulate_g2");
  {
    const unsigned cblocks = (unsigned)(((size_t)tv.NB * (size_t)tv.combine_lanes + 63) / 64);
    if (tv.combine_lanes == 1) hipLaunchKernelGGL(k2_combine<1>, dim3(cblocks), dim3(64), 0, stream, tv.task_off, tv.NB, tv.partials, tv.pyrA, tv.seq_parts);
    else if (tv.combine_lanes == 2) hipLaunchKernelGGL(k2_combine<2>, dim3(cblocks), dim3(64), 0, stream, tv.task_off, tv.NB, tv.partials, tv.pyrA, tv.seq_parts);
    else if (tv.combine_lanes == 4) hipLaunchKernelGGL(k2_combine<4>, dim3(cblocks), dim3(64), 0, stream, tv.task_off, tv.NB, tv.partials, tv.pyrA, tv.seq_parts);
    else hipLaunchKernelGGL(k2_combine<8>, dim3(cblocks), dim3(64), 0, stream, tv.task_off, tv.NB, tv.partials, tv.pyrA, tv.seq_parts);
  }
  hipLaunchKernelGGL(k2_combine_heavy, dim3(256), dim3(64), 0, stream, tv.task_off, tv.partials, tv.pyrA, tv.heavy_count, (const uint2*)tv.heavy, tv.heavy_cap);
  prof_mark(stream, "combine_g2");
  uint32_t* cur = tv.pyrA;
  uint32_t* nxt = tv.pyrB;
  uint32_t in_stride = tv.B, N = tv.B;
  int s = 1;
  while (N > 2) {
    const uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
    if ((size_t)per_win * tv.WB <= 32768) hipLaunchKernelGGL(k2_pyramid_step_quad, dim3((4 * per_win + 63) / 64, tv.WB), dim3(64), 0, stream, cur, nxt, N, s, in_stride, per_win);
    else hipLaunchKernelGGL(k2_pyramid_step, dim3((per_win + 63) / 64, tv.WB), dim3(64), 0, stream, cur, nxt, N, s, in_stride, per_win);
    uint32_t* t = cur; cur = nxt; nxt = t;
    in_stride = per_win;
    N >>= 1;
    s++;
  }
  const int nz = s - 1;
  prof_mark(stream, "pyramid_g2");
  hipLaunchKernelGGL(k2_window_sum, dim3(tv.WB), dim3(128), 0, stream, cur, in_stride, nz, tv.winsum, tv.WB);
  hipLaunchKernelGGL(k2_fold, dim3(1), dim3(64), 0, stream, tv.winsum, tv.WB, c, d_out);
  prof_mark(stream, "fold_g2");
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

int test_g2_op(int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (op >= 3) hipLaunchKernelGGL(k2_test_op_quad, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, stream, op, d_a, d_b, d_out, n);
  else hipLaunchKernelGGL(k2_test_op, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, op, d_a, d_b, d_out, n);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

}  // namespace zkhip
