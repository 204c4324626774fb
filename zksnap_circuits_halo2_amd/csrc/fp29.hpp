// 254-bit prime-field arithmetic for gfx950, radix 2^29, 9 limbs, Montgomery radix 2^261.
//
// Why this shape (measured on MI355X, profiles/r01_isa_issue_rates.txt): v_mad_u64_u32 issues at
// ~5.6 cycles per wave-instruction, the same as an add-with-carry (4.9) and v_fma_f64 (5.2), so the
// cost of a modular multiply is its *instruction count*.  With 29-bit limbs every partial product
// is < 2^58 and a whole product-scanning column (9 a_i*b_j + 9 m_i*p_j) fits a 64-bit accumulator:
// one v_mad_u64_u32 per partial product, no carry chain, and field add/sub are 9 plain 32-bit adds.
// The 7 spare bits (2^261 / p ~ 169) let point formulas run without conditional subtractions.
//
// Replaces the 4x64 Montgomery arithmetic of halo2curves bn256::{Fq,Fr} [DEP] (types reached from
// /root/reference/aggregator/src/wrapper.rs:8); external memory format stays 4 x u64 LE limbs,
// Montgomery radix 2^256 -- see load_ext / store_ext.
//
// Magnitude discipline (checked by the comments at every call site):
//   N   : "normalised" -- limbs 0..7 < 2^29, limb 8 holds the rest.  Every fe_mul/fe_sqr output is N
//         and has value < p * (a*b / (p * 2^261) + 1)  (< 2p whenever a*b < 169 p^2).
//   fe_mul(a, b) needs  9 * max(a_i) * max(b_j) + 9 * 2^58 + 2^35 < 2^64, i.e. max(a_i)*max(b_j) < 2^60.6:
//         N x (limbs < 2^31.5)  or  (limbs < 2^30.3) x (limbs < 2^30.3).
//   fe_sub(a, b, K) = a + K*p - b limb-wise, K*p in the borrow-proof form RED(K): needs b in N form
//         with value < (K-1) p.
#pragma once
#include <cstdint>
#include "bn254_constants.hpp"
#if defined(__HIPCC__)
#include <type_traits>
#include "mac_blocks.hpp"
#endif

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#else
#define ZK_HD inline
#define ZK_D inline
#endif

namespace zkhip {

constexpr int NL = 9;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1;

struct fe {
  uint32_t l[NL];
};

ZK_HD fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = 0;
  return r;
}

template <class P>
ZK_HD fe fe_const(const uint32_t (&c)[NL]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c[i];
  return r;
}

template <class P> ZK_HD fe fe_one() { fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = P::ONE[i];
  return r; }

// acc += x * y.  CHAIN keeps a product column one chain of v_mad_u64_u32 whose addend is the running sum.  Left alone, LLVM
// reassociates each column into partial sums that start from 0 and joins them (and the carry of the previous column) with
// v_lshl_add_u64: 17-34 extra instructions per product.  The two shapes suit different kernels (measured):
// the chain wins where many waves per SIMD hide its latency and instruction issue is the bound (bucket accumulation: -5 %); the
// compiler's shape wins in the latency-bound tail and, by a wide margin, in the NTT, where independent butterflies want to
// interleave.  On the device the chain is written as asm blocks of whole column runs (fe_mul_blocks below, mac_blocks.hpp); this
// per-product pin (an empty asm that emits nothing but makes the value opaque) is its portable statement and what the host pass sees.
template <bool CHAIN>
ZK_HD void fe_mac(uint64_t& acc, uint32_t x, uint32_t y) {
  acc += (uint64_t)x * y;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CHAIN) asm("" : "+v"(acc));
#endif
}

#if defined(__HIPCC__)
// compile-time loop: f(std::integral_constant<int, K>) for K = BEGIN .. END - 1 (the block sizes below are template arguments)
template <int K, int END, class F>
ZK_D void static_for(F&& f) {
  if constexpr (K < END) {
    f(std::integral_constant<int, K>{});
    static_for<K + 1, END>(f);
  }
}

// The CHAIN shape of fe_mul / fe_sqr / fe_mul_add with each run of multiply-adds of a column as one asm block (mac_blocks.hpp):
// same values as the plain code, column by column.  `second`: an optional second product c*d under the same reduction.
template <class P, bool SECOND>
ZK_D fe fe_mul_blocks(const fe& a, const fe& b, const fe& c, const fe& d) {
  uint64_t acc = 0;
  uint32_t m[NL];
  fe r;
  static_for<0, NL>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    uint32_t x[NL], y[NL];
#pragma unroll
    for (int i = 0; i <= k; i++) { x[i] = a.l[i]; y[i] = b.l[k - i]; }
    mac_vv<k + 1>(acc, x, y);
    if constexpr (SECOND) {
#pragma unroll
      for (int i = 0; i <= k; i++) { x[i] = c.l[i]; y[i] = d.l[k - i]; }
      mac_vv<k + 1>(acc, x, y);
    }
    if constexpr (k > 0) {
#pragma unroll
      for (int i = 0; i < k; i++) { x[i] = m[i]; y[i] = P::P[k - i]; }
      mac_vs<k>(acc, x, y);
    }
    m[k] = ((uint32_t)acc * P::INV) & LMASK;
    acc += (uint64_t)m[k] * P::P[0];
    acc >>= LB;
  });
  static_for<NL, 2 * NL - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value, lo = k - NL + 1, n = NL - lo;
    uint32_t x[NL], y[NL];
#pragma unroll
    for (int i = 0; i < n; i++) { x[i] = a.l[lo + i]; y[i] = b.l[k - lo - i]; }
    mac_vv<n>(acc, x, y);
    if constexpr (SECOND) {
#pragma unroll
      for (int i = 0; i < n; i++) { x[i] = c.l[lo + i]; y[i] = d.l[k - lo - i]; }
      mac_vv<n>(acc, x, y);
    }
#pragma unroll
    for (int i = 0; i < n; i++) { x[i] = m[lo + i]; y[i] = P::P[k - lo - i]; }
    mac_vs<n>(acc, x, y);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  });
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

template <class P>
ZK_D fe fe_sqr_blocks(const fe& a) {
  uint64_t acc = 0;
  uint32_t m[NL], dbl[NL];
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) dbl[i] = a.l[i] << 1;
  static_for<0, 2 * NL - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value, lo = k < NL ? 0 : k - NL + 1;
    constexpr int cross = (k + 1) / 2 - lo;                  // products 2 a_i a_(k-i), lo <= i, 2 i < k
    uint32_t x[NL], y[NL];
    if constexpr (cross > 0) {
#pragma unroll
      for (int i = 0; i < cross; i++) { x[i] = dbl[lo + i]; y[i] = a.l[k - lo - i]; }
      mac_vv<cross>(acc, x, y);
    }
    if constexpr ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
    constexpr int nm = k < NL ? k : NL - lo;                 // m_i p_(k-i): i < k (first half), lo <= i < NL (second half)
    if constexpr (nm > 0) {
#pragma unroll
      for (int i = 0; i < nm; i++) { x[i] = m[lo + i]; y[i] = P::P[k - lo - i]; }
      mac_vs<nm>(acc, x, y);
    }
    if constexpr (k < NL) {
      m[k] = ((uint32_t)acc * P::INV) & LMASK;
      acc += (uint64_t)m[k] * P::P[0];
    } else {
      r.l[k - NL] = (uint32_t)acc & LMASK;
    }
    acc >>= LB;
  });
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}
#endif

// Montgomery product a*b*2^-261 mod p (lazy: result in N form, value < p*(ab/(p 2^261) + 1)).
template <class P, bool CHAIN = false>
ZK_HD fe fe_mul(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CHAIN) return fe_mul_blocks<P, false>(a, b, a, b);
#endif
  uint64_t acc = 0;
  uint32_t m[NL];
  fe r;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) fe_mac<CHAIN>(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    m[k] = ((uint32_t)acc * P::INV) & LMASK;
    fe_mac<CHAIN>(acc, m[k], P::P[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// (a*b + c*d) * 2^-261 mod p with ONE reduction: the two product columns share the accumulator, so the 81 multiply-adds (and
// the carry handling) of a second Montgomery reduction disappear.  Used for Y3 = R*T - PPP*Y1 with d = K p - Y1.
// Needs 9 * (max a_i * max b_j + max c_i * max d_j) + 9 * 2^58 + 2^35 < 2^64; value < p * ((ab + cd) / (p 2^261) + 1).
template <class P, bool CHAIN = false>
ZK_HD fe fe_mul_add(const fe& a, const fe& b, const fe& c, const fe& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CHAIN) return fe_mul_blocks<P, true>(a, b, c, d);
#endif
  uint64_t acc = 0;
  uint32_t m[NL];
  fe r;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) fe_mac<CHAIN>(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i <= k; i++) fe_mac<CHAIN>(acc, c.l[i], d.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    m[k] = ((uint32_t)acc * P::INV) & LMASK;
    fe_mac<CHAIN>(acc, m[k], P::P[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, c.l[i], d.l[k - i]);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Montgomery square (45 distinct products instead of 81).  Needs limbs < 2^30.3.
template <class P, bool CHAIN = false>
ZK_HD fe fe_sqr(const fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CHAIN) return fe_sqr_blocks<P>(a);
#endif
  uint64_t acc = 0;
  uint32_t m[NL], d[NL];
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) d[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < NL; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) fe_mac<CHAIN>(acc, d[i], a.l[k - i]);
    if ((k & 1) == 0) fe_mac<CHAIN>(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
    for (int i = 0; i < k; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    m[k] = ((uint32_t)acc * P::INV) & LMASK;
    fe_mac<CHAIN>(acc, m[k], P::P[0]);
    acc >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; i++) fe_mac<CHAIN>(acc, d[i], a.l[k - i]);
    if ((k & 1) == 0) fe_mac<CHAIN>(acc, a.l[k / 2], a.l[k / 2]);
#pragma unroll
    for (int i = k - NL + 1; i < NL; i++) fe_mac<CHAIN>(acc, m[i], P::P[k - i]);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

ZK_HD fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}

ZK_HD fe fe_dbl(const fe& a) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
  return r;
}

// a + RED - b, limb-wise; RED is a borrow-proof multiple of p (see header).
ZK_HD fe fe_sub_red(const fe& a, const fe& b, const uint32_t (&red)[NL]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + red[i] - b.l[i];
  return r;
}

// RED - b
ZK_HD fe fe_neg_red(const fe& b, const uint32_t (&red)[NL]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = red[i] - b.l[i];
  return r;
}

// Carry-propagate to N form (value unchanged).
ZK_HD fe fe_norm(const fe& a) {
  fe r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    uint32_t t = a.l[i] + c;
    r.l[i] = t & LMASK;
    c = t >> LB;
  }
  r.l[NL - 1] = a.l[NL - 1] + c;
  return r;
}

ZK_HD bool fe_is_zero_limbs(const fe& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) o |= a.l[i];
  return o == 0;
}

// For an N-form value < 2p (a multiply output): is it 0 mod p?
template <class P>
ZK_HD bool fe_mulout_is_zero(const fe& a) {
  uint32_t o = 0, e = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    o |= a.l[i];
    e |= a.l[i] ^ P::P[i];
  }
  return (o == 0) | (e == 0);
}

// N-form value < 2p  ->  canonical [0, p)
template <class P>
ZK_HD fe fe_canon_lt2p(const fe& a) {
  fe d;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    int32_t t = (int32_t)a.l[i] - (int32_t)P::P[i] + borrow;
    borrow = t >> 31;  // 0 or -1
    d.l[i] = (uint32_t)t & LMASK;
  }
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
  return r;
}

// Cheap partial reduction: N-form value < 2^261  ->  N-form value < 2p + 2^233 (< 2^256), same residue.
// q = floor(top_limb / (p_top + 1)) never exceeds floor(x / p) (so x - q p >= 0) and undershoots it by < 2, where
// p_top = p >> 232 (0x30644e for both BN254 fields).  ~50 instructions instead of a 215-instruction multiply.
template <class P>
ZK_HD fe fe_reduce_soft(const fe& x) {
  constexpr uint64_t MAGIC = (1ull << 52) / (uint64_t)(P::P[NL - 1] + 1);
  const uint32_t q = (uint32_t)(((uint64_t)x.l[NL - 1] * MAGIC) >> 52);
  fe r;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    int64_t v = (int64_t)x.l[i] - (int64_t)((uint64_t)q * P::P[i]) + carry;
    r.l[i] = (uint32_t)v & LMASK;
    carry = v >> LB;
  }
  r.l[NL - 1] = (uint32_t)((int64_t)x.l[NL - 1] - (int64_t)((uint64_t)q * P::P[NL - 1]) + carry);
  return r;
}

// N-form value < 3p  ->  canonical [0, p)  (two conditional subtractions)
template <class P>
ZK_HD fe fe_canon_lt3p(const fe& a) { return fe_canon_lt2p<P>(fe_canon_lt2p<P>(a)); }

// Any lazily-reduced value (< 2^261, limbs < 2^31.5) -> canonical [0,p), same Montgomery form.
template <class P>
ZK_HD fe fe_canon(const fe& a) {
  fe one = fe_one<P>();
  return fe_canon_lt2p<P>(fe_mul<P>(one, a));  // a * 2^261 * 2^-261 = a, value < ~1.1 p
}

// ---- external format: 8 x u32 little-endian words of a 256-bit integer -----------------------
// unpack with a left shift of SH bits (SH = 5 turns x*2^256 into the integer x*2^261, unreduced < 32p)
template <int SH>
ZK_HD fe fe_unpack(const uint32_t (&w)[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int s = 32 * j + SH - LB * i;  // bit position of word j relative to limb i
      if (s > -32 && s < LB) v |= (s >= 0) ? (w[j] << (s & 31)) : (w[j] >> ((-s) & 31));
    }
    r.l[i] = v & LMASK;
  }
  if (SH + 256 > LB * (NL - 1) + LB) {}  // (values always fit: 256 + 5 = 261)
  return r;
}

// pack an N-form value < 2^256 into 8 words
ZK_HD void fe_pack(const fe& a, uint32_t (&w)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int s = LB * i - 32 * j;
      if (s > -LB && s < 32) v |= (s >= 0) ? (a.l[i] << (s & 31)) : (a.l[i] >> ((-s) & 31));
    }
    w[j] = v;
  }
}

#if defined(__HIPCC__)
// 32-byte element loads/stores as two 16-byte vector accesses (global_load_dwordx4).
ZK_D void load_words(const uint32_t* p, uint32_t (&w)[8]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
ZK_D void store_words(uint32_t* p, const uint32_t (&w)[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
#endif

// external Montgomery-256 element -> internal Montgomery-261, unreduced (< 32p), N-limbed.  Free.
ZK_HD fe fe_from_ext_lazy(const uint32_t (&w)[8]) { return fe_unpack<5>(w); }

// internal (lazy) -> canonical external Montgomery-256 words
template <class P>
ZK_HD void fe_to_ext(const fe& a, uint32_t (&w)[8]) {
  fe c;
#pragma unroll
  for (int i = 0; i < NL; i++) c.l[i] = P::TO_EXT[i];
  fe_pack(fe_canon_lt2p<P>(fe_mul<P>(c, a)), w);  // a*2^261 * 2^256 * 2^-261
}

}  // namespace zkhip
