// Field inversion by "safegcd" division steps (D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular
// inversion", TCHES 2019: the published algorithm; the 30-bit batching of the transition matrices follows the form that paper's
// section 11 and the public write-ups of it describe).  Replaces the Fermat chain a^(p-2) (254 squarings + ~127 multiplications,
// ~82 k instructions of ONE thread: 0.34 ms of latency under every batch inversion, grand product and affine normalisation) with
// 20 rounds of 30 division steps: ~11 k instructions, no data-dependent branch (every lane of a wavefront runs the same trip counts).
//
// Where the reference does this: `Field::invert` / `BatchInvert::batch_invert` [DEP ff / halo2curves-axiom, un-vendored], under the
// permutation / lookup grand products and `Curve::batch_normalize` of create_proof (/root/reference/aggregator/src/wrapper.rs:129).
// An inverse is unique, so the canonical result equals the Fermat chain's bit for bit (checked in tests/cpp/fp29_host_check.cpp
// against an independent big-integer a^(p-2), and on the GPU through every batch-inversion / normalisation parity test).
//
// Shape: f = p, g = x (plain integers, 9 signed limbs of 30 bits), d = 0, e = 1.  Each round runs 30 division steps on the low
// 30 bits of (f, g) alone, which yields a 2x2 integer matrix t with t * (f, g) = 2^30 * (f', g'); the matrix is then applied once to
// the full-width (f, g) and, modulo p, to (d, e).  After 600 steps (590 suffice for 256-bit inputs) g = 0, f = +-1 and d = +-x^-1.
#pragma once
#include "fp29.hpp"

namespace zkhip {

namespace inv30 {

constexpr int NS = 9;                          // 9 x 30 bits = 270 >= 256
constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);

struct s30 { int32_t v[NS]; };
struct mat { int32_t u, v, q, r; };

// limb i (30 bits) of the integer held in 29-bit limbs c[0..8]
template <class A>
ZK_HD constexpr uint32_t gather30(const A& c, int i) {
  uint32_t out = 0;
  for (int j = 0; j < NL; j++) {
    const int s = LB * j - 30 * i;             // position of 29-bit limb j relative to 30-bit limb i
    if (s > -LB && s < 30) out |= (s >= 0) ? ((uint32_t)c[j] << s) : ((uint32_t)c[j] >> (-s));
  }
  return out & (uint32_t)M30;
}

template <class P> ZK_HD constexpr int32_t modulus_limb(int i) { return (int32_t)gather30(P::P, i); }

// p^-1 mod 2^30 (Newton iteration on the low limb)
template <class P> ZK_HD constexpr uint32_t modulus_inv30() {
  const uint32_t p0 = gather30(P::P, 0) | (gather30(P::P, 1) << 30);
  uint32_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
  return x & (uint32_t)M30;
}

// 30 division steps on the low bits; zeta = -(delta + 1/2).  Returns the new zeta and the transition matrix.
ZK_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, mat& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 6
#endif
  for (int i = 0; i < 30; i++) {
    const uint32_t neg = (uint32_t)(zeta >> 31);          // all ones when zeta < 0
    const uint32_t odd = 0u - (g & 1u);                   // all ones when g is odd
    const uint32_t x = (f ^ neg) - neg, y = (u ^ neg) - neg, z = (v ^ neg) - neg;   // (f, u, v) negated when zeta < 0
    g += x & odd; q += y & odd; r += z & odd;
    const uint32_t swap = neg & odd;
    zeta = (zeta ^ (int32_t)swap) - 1;                    // -zeta - 2 on a swap, zeta - 1 otherwise
    f += g & swap; u += q & swap; v += r & swap;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return zeta;
}

// (f, g) <- t * (f, g) / 2^30 (exact)
ZK_HD void update_fg(s30& f, s30& g, const mat& t) {
  int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0];
  int64_t cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < NS; i++) {
    cf += (int64_t)t.u * f.v[i] + (int64_t)t.v * g.v[i];
    cg += (int64_t)t.q * f.v[i] + (int64_t)t.r * g.v[i];
    f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
    g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
  }
  f.v[NS - 1] = (int32_t)cf;
  g.v[NS - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod p; d, e stay in (-2p, p)
template <class P>
ZK_HD void update_de(s30& d, s30& e, const mat& t) {
  const int32_t sd = d.v[NS - 1] >> 31, se = e.v[NS - 1] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);      // + p per negative input row
  int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0];
  int64_t ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
  constexpr uint32_t PINV = modulus_inv30<P>();
  md -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);    // multiples of p that clear the low 30 bits
  me -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)modulus_limb<P>(0) * md;
  ce += (int64_t)modulus_limb<P>(0) * me;
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < NS; i++) {
    cd += (int64_t)t.u * d.v[i] + (int64_t)t.v * e.v[i] + (int64_t)modulus_limb<P>(i) * md;
    ce += (int64_t)t.q * d.v[i] + (int64_t)t.r * e.v[i] + (int64_t)modulus_limb<P>(i) * me;
    d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
    e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
  }
  d.v[NS - 1] = (int32_t)cd;
  e.v[NS - 1] = (int32_t)ce;
}

// d in (-2p, p), negated when sign < 0  ->  [0, p)
template <class P>
ZK_HD void normalize(s30& r, int32_t sign) {
  const int32_t add1 = r.v[NS - 1] >> 31, neg = sign >> 31;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    int32_t x = r.v[i] + (modulus_limb<P>(i) & add1);
    r.v[i] = (x ^ neg) - neg;
  }
#pragma unroll
  for (int i = 0; i < NS - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
  const int32_t add2 = r.v[NS - 1] >> 31;
#pragma unroll
  for (int i = 0; i < NS; i++) r.v[i] += modulus_limb<P>(i) & add2;
#pragma unroll
  for (int i = 0; i < NS - 1; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
}

}  // namespace inv30

// x^-1 mod p of the plain integer held in `x` (canonical: N form, value < p); returns the plain integer inverse, canonical.  0 -> 0.
template <class P>
ZK_HD fe fe_inverse_plain(const fe& x) {
  using namespace inv30;
  s30 d, e, f, g;
#pragma unroll
  for (int i = 0; i < NS; i++) {
    d.v[i] = 0; e.v[i] = i == 0;
    f.v[i] = modulus_limb<P>(i);
    g.v[i] = (int32_t)gather30(x.l, i);
  }
  int32_t zeta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int it = 0; it < 20; it++) {
    mat t;
    zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de<P>(d, e, t);
    update_fg(f, g, t);
  }
  normalize<P>(d, f.v[NS - 1]);
  fe r;                                         // 30-bit limbs -> 29-bit limbs
#pragma unroll
  for (int i = 0; i < NL; i++) {
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const int s = 30 * j - LB * i;
      if (s > -30 && s < LB) out |= (s >= 0) ? ((uint32_t)d.v[j] << s) : ((uint32_t)d.v[j] >> (-s));
    }
    r.l[i] = out & LMASK;
  }
  return r;
}

// Montgomery-261 inverse: a = x * 2^261 (any lazily reduced value the multiplication accepts)  ->  x^-1 * 2^261, N form, < 2p.  0 -> 0.
template <class P>
ZK_HD fe fe_inverse(const fe& a) {
  const fe plain = fe_inverse_plain<P>(fe_canon<P>(a));          // (x 2^261)^-1 as a plain integer
  fe r3;
#pragma unroll
  for (int i = 0; i < NL; i++) r3.l[i] = P::R3[i];
  return fe_mul<P>(plain, r3);                                    // x^-1 2^-261 * 2^783 * 2^-261 = x^-1 2^261
}

}  // namespace zkhip
