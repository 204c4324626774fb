// Elementwise field / curve kernels behind the zkhip_test_* parity hooks:
// they expose the device arithmetic of fp29.hpp / ec.hpp in the external memory format so tests can diff it
// against the oracle.  Not used by the MSM / NTT paths themselves.
#include <hip/hip_runtime.h>
#include "ec_quad.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

template <class P>
__global__ void __launch_bounds__(256) k_field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[8], wb[8], wo[8];
  load_words(a + i * 8, wa);
  load_words(b + i * 8, wb);
  fe one = fe_one<P>();
  fe x = fe_mul<P>(one, fe_from_ext_lazy(wa));   // x * 2^261, reduced
  fe y = fe_mul<P>(one, fe_from_ext_lazy(wb));
  fe r;
  switch (op) {
    case 0: r = fe_mul<P>(x, y); break;
    case 1: r = fe_add(x, y); break;
    case 2: r = fe_sub_red(x, y, P::P3_S1); break;
    default: r = fe_sqr<P>(x); break;
  }
  fe_to_ext<P>(r, wo);
  store_words(out + i * 8, wo);
}

__global__ void __launch_bounds__(128) k_g1_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine_words pa = load_affine(a, i), pb = load_affine(b, i);
  xyzz acc = xyzz_identity();
  if (!affine_is_identity(pa)) xyzz_madd(acc, fe_from_ext_lazy(pa.x), fe_from_ext_lazy(pa.y));
  if (op == 1) {
    acc = xyzz_dbl(acc);
  } else if (!affine_is_identity(pb)) {
    fe y2 = fe_from_ext_lazy(pb.y);
    if (op == 2) y2 = fe_neg_red(y2, Fq::P64_S1);
    xyzz_madd(acc, fe_from_ext_lazy(pb.x), y2);
  }
  store_jacobian(acc, out + i * 24);
}

// quad-cooperative formulas (ec_quad.hpp), 4 lanes per element: op 3 = 2a + 2b, op 4 = 4a
__global__ void __launch_bounds__(128) k_g1_op_quad(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gid >> 2;
  const uint32_t q = (uint32_t)gid & 3;
  if (i >= n) return;                                   // whole quads leave together
  affine_words pa = load_affine(a, i), pb = load_affine(b, i);
  xyzz A = xyzz_identity(), B = xyzz_identity();
  if (!affine_is_identity(pa)) xyzz_madd(A, fe_from_ext_lazy(pa.x), fe_from_ext_lazy(pa.y));
  if (!affine_is_identity(pb)) xyzz_madd(B, fe_from_ext_lazy(pb.x), fe_from_ext_lazy(pb.y));
  xyzz r;
  if (op == 3) r = xyzz_add_quad(xyzz_dbl_quad(A, q), xyzz_dbl_quad(B, q), q);
  else r = xyzz_dbl_quad(xyzz_dbl_quad(A, q), q);
  if (q == 0) store_jacobian(r, out + i * 24);
}

// `G1Affine::read_raw` check of halo2curves [DEP] (what `SerdeFormat::RawBytes` readers run on every point of an SRS or key file):
// both coordinates canonical Montgomery residues (< q) and (x, y) = (0, 0) or y^2 = x^3 + 3.  first_bad receives the smallest
// index that fails (stays n when every point passes).
__device__ __forceinline__ bool words_below_q(const uint32_t (&w)[8]) {
  const fe v = fe_unpack<0>(w);                   // the raw 256-bit integer as limbs
  bool lt = false, eq = true;
#pragma unroll
  for (int i = NL - 1; i >= 0; i--) {
    lt = lt || (eq && v.l[i] < Fq::P[i]);
    eq = eq && v.l[i] == Fq::P[i];
  }
  return lt;
}

__global__ void __launch_bounds__(256) k_g1_check(const uint32_t* __restrict__ pts, size_t n, unsigned long long* __restrict__ first_bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine_words p = load_affine(pts, i);
  if (affine_is_identity(p)) return;
  bool ok = words_below_q(p.x) && words_below_q(p.y);
  if (ok) {
    const fe one = fe_one<Fq>();
    const fe x = fe_mul<Fq>(one, fe_from_ext_lazy(p.x)), y = fe_mul<Fq>(one, fe_from_ext_lazy(p.y));   // internal form, < 2p
    const fe lhs = fe_sqr<Fq>(y);
    const fe x3 = fe_mul<Fq>(fe_sqr<Fq>(x), x);
    const fe three = fe_norm(fe_add(fe_add(one, one), one));                   // 3 in internal form, < 3p
    const fe rhs = fe_norm(fe_add(x3, three));                                 // < 5p, N form
    const fe diff = fe_mul<Fq>(one, fe_sub_red(lhs, rhs, Fq::P6_S1));          // same residue, < 2p
    ok = fe_mulout_is_zero<Fq>(diff);
  }
  if (!ok) atomicMin(first_bad, (unsigned long long)i);
}

int g1_check_points_device(const uint32_t* d_points, size_t n, unsigned long long* d_first_bad, hipStream_t stream) {
  const unsigned long long init = (unsigned long long)n;
  if (hipMemcpyAsync(d_first_bad, &init, 8, hipMemcpyHostToDevice, stream) != hipSuccess) { set_error("g1_check: upload failed"); return ZKHIP_EHIP; }
  if (hipStreamSynchronize(stream) != hipSuccess) return ZKHIP_EHIP;      // `init` is a stack variable
  if (n) hipLaunchKernelGGL(k_g1_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_points, n, d_first_bad);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

int test_field_op(int field, int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (field == 0) hipLaunchKernelGGL(k_field_op<FqParams>, grid, block, 0, stream, op, d_a, d_b, d_out, n);
  else hipLaunchKernelGGL(k_field_op<FrParams>, grid, block, 0, stream, op, d_a, d_b, d_out, n);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

int test_g1_op(int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (op >= 3) hipLaunchKernelGGL(k_g1_op_quad, dim3((unsigned)((4 * n + 127) / 128)), dim3(128), 0, stream, op, d_a, d_b, d_out, n);
  else hipLaunchKernelGGL(k_g1_op, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, stream, op, d_a, d_b, d_out, n);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

}  // namespace zkhip
