// `SerdeFormat::Processed` point encoding: G1Affine <-> 32 bytes, batched on the device.
//
// Replaces `GroupEncoding::{to_bytes, from_bytes}` of halo2curves bn256::G1Affine [DEP halo2curves `derive/curve.rs`], which
// `ParamsKZG::{write_custom, read_custom}`, `VerifyingKey` / `ProvingKey::{write, read}` call once per point under
// `SerdeFormat::Processed` (the reference itself writes RawBytesUnchecked: /root/reference/aggregator/src/wrapper.rs:971-988).
// Restated from the published encoding -- the reference tree holds no compressed file, so the flag layout is unpinned:
//   layout 0 (halo2curves >= 0.3.2): x canonical little-endian, bit 6 of the last byte = lsb of canonical y, bit 7 = identity (x = 0);
//   layout 1 (earlier releases):     bit 7 of the last byte = lsb of canonical y, the identity is 32 zero bytes.
// Decoding is strict: x must be canonical (< q), x^3 + 3 must be a square, and an identity flag with x != 0 is an error.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "ec.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

using Fq = FqParams;

// (q + 1) / 4: q = 3 mod 4, so a square a has the roots +-a^((q + 1) / 4)
__constant__ uint32_t SQRT_EXP[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
constexpr int SQRT_EXP_BITS = 252;

__device__ __forceinline__ fe fq_const(const uint32_t (&c)[9]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c[i];
  return r;
}

__device__ __forceinline__ bool raw_below_q(const fe& v) {      // v = limbs of a raw 256-bit integer
  bool lt = false, eq = true;
#pragma unroll
  for (int i = NL - 1; i >= 0; i--) {
    lt = lt || (eq && v.l[i] < Fq::P[i]);
    eq = eq && v.l[i] == Fq::P[i];
  }
  return lt;
}

// internal form (any multiply output) -> the canonical integer's limbs
__device__ __forceinline__ fe fq_plain(const fe& a) { return fe_canon_lt2p<Fq>(fe_mul<Fq>(fq_const(Fq::RAW_ONE), a)); }

__global__ void __launch_bounds__(256) k_g1_compress(const uint32_t* __restrict__ pts, size_t n, uint32_t* __restrict__ out, int layout) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const affine_words p = load_affine(pts, i);
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (affine_is_identity(p)) {
    if (layout == 0) w[7] = 0x80000000u;
  } else {
    const fe from_ext = fq_const(Fq::FROM_EXT_CANON);
    const fe x = fe_canon_lt2p<Fq>(fe_mul<Fq>(from_ext, fe_unpack<0>(p.x)));     // x * 2^256 -> x
    const fe y = fe_canon_lt2p<Fq>(fe_mul<Fq>(from_ext, fe_unpack<0>(p.y)));
    fe_pack(x, w);
    w[7] |= (y.l[0] & 1u) << (layout == 0 ? 30 : 31);
  }
  store_words(out + i * 8, w);
}

__global__ void __launch_bounds__(256) k_g1_decompress(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ pts, int layout,
                                                       unsigned long long* __restrict__ first_bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words(in + i * 8, w);
  bool is_inf, sign;
  if (layout == 0) {
    is_inf = (w[7] >> 31) != 0;
    sign = ((w[7] >> 30) & 1u) != 0;
    w[7] &= 0x3fffffffu;
  } else {
    sign = (w[7] >> 31) != 0;
    w[7] &= 0x7fffffffu;
    is_inf = !sign && (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0;
  }
  uint32_t ox[8] = {0, 0, 0, 0, 0, 0, 0, 0}, oy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool ok = true;
  const fe xr = fe_unpack<0>(w);
  if (is_inf) {
    ok = (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0 && !sign;
  } else if (!raw_below_q(xr)) {
    ok = false;
  } else {
    const fe one = fe_one<Fq>();
    const fe x = fe_mul<Fq>(fq_const(Fq::R2), xr);                               // x * 2^261, < 2p
    const fe three = fe_norm(fe_add(fe_add(one, one), one));
    const fe y2 = fe_mul<Fq>(one, fe_norm(fe_add(fe_mul<Fq>(fe_sqr<Fq>(x), x), three)));   // x^3 + 3, a multiply output again
    fe y = one;
#pragma unroll 1
    for (int b = SQRT_EXP_BITS - 1; b >= 0; b--) {                               // left-to-right square and multiply (the exponent is public)
      y = fe_sqr<Fq>(y);
      if ((SQRT_EXP[b >> 5] >> (b & 31)) & 1u) y = fe_mul<Fq>(y, y2);
    }
    const fe diff = fe_mul<Fq>(one, fe_sub_red(fe_sqr<Fq>(y), y2, Fq::P3_S1));   // y^2 - (x^3 + 3)
    ok = fe_mulout_is_zero<Fq>(diff);
    if (ok) {
      const fe yp = fq_plain(y);
      if (((yp.l[0] & 1u) != 0) != sign) y = fe_norm(fe_neg_red(y, Fq::P2_S1));  // the other root: 2p - y
      fe_to_ext<Fq>(x, ox);
      fe_to_ext<Fq>(y, oy);
    }
  }
  if (!ok) atomicMin(first_bad, (unsigned long long)i);
  uint4* q = reinterpret_cast<uint4*>(pts + i * 16);
  q[0] = make_uint4(ox[0], ox[1], ox[2], ox[3]);
  q[1] = make_uint4(ox[4], ox[5], ox[6], ox[7]);
  q[2] = make_uint4(oy[0], oy[1], oy[2], oy[3]);
  q[3] = make_uint4(oy[4], oy[5], oy[6], oy[7]);
}

int g1_compress_device(const uint32_t* d_points, size_t n, uint32_t* d_out, int layout, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_g1_compress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_points, n, d_out, layout);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

// d_first_bad must hold n before the launch (the caller initialises it on the stream)
int g1_decompress_device(const uint32_t* d_in, size_t n, uint32_t* d_points, int layout, unsigned long long* d_first_bad, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_g1_decompress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_in, n, d_points, layout, d_first_bad);
  return hipGetLastError() == hipSuccess ? ZKHIP_OK : ZKHIP_EHIP;
}

}  // namespace zkhip
