// Device helpers shared by the Fr-vector kernels (poly.hip, rowvm.hip): constants passed by value, external-domain loads
// and stores, small-exponent powers.  External domain = the caller's 4 x u64 Montgomery-256 words (fp29.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "fp29.hpp"

namespace zkhip {

using Fr = FrParams;

struct fe_arg {   // a field constant passed by value (external words)
  uint32_t w[8];
};

__device__ __forceinline__ fe fr_const_internal(const fe_arg& c) {   // c * 2^256 -> c * 2^261, reduced
  fe k;
#pragma unroll
  for (int i = 0; i < NL; i++) k.l[i] = Fr::FROM_EXT[i];
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = c.w[i];
  return fe_mul<Fr>(k, fe_unpack<0>(w));
}

__device__ __forceinline__ fe load_ext(const uint32_t* p, size_t i) {
  uint32_t w[8];
  load_words(p + i * 8, w);
  return fe_unpack<0>(w);
}
__device__ __forceinline__ void store_canon(uint32_t* p, size_t i, const fe& x_lt3p) {
  uint32_t w[8];
  fe_pack(fe_canon_lt3p<Fr>(x_lt3p), w);
  store_words(p + i * 8, w);
}

__device__ __forceinline__ fe fr_pow_u32(fe base, uint32_t e) {   // base^e, base internal & reduced
  fe acc = fe_one<Fr>();
  while (e) {
    if (e & 1) acc = fe_mul<Fr>(acc, base);
    base = fe_sqr<Fr>(base);
    e >>= 1;
  }
  return acc;
}

}  // namespace zkhip
