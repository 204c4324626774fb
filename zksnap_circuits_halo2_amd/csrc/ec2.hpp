// BN254 G2: the quadratic extension Fq2 = Fq[u] / (u^2 + 1) over the lazy radix-2^29 field of fp29.hpp, and extended Jacobian
// "XYZZ" point arithmetic on the sextic twist y^2 = x^3 + 3 / (9 + u) over it.
//
// Replaces halo2curves bn256::{Fq2, G2Affine, G2} [DEP] for `best_multiexp::<G2Affine>`.  No call site in the reference prover reaches an
// MSM over G2 (it only reads `params.g2()` / `params.s_g2()`: /root/reference/aggregator/src/wrapper.rs:1142-1144); the north star names
// "MSM on BN254 G1/G2", so the path exists, written for correctness first: unlike ec.hpp there is no bound tracking across operations.
// Every Fq2 operation returns both components in STANDARD FORM -- N form (fp29.hpp), value < 2p + 2^233 -- by ending with the ~50
// instruction quotient-estimate reduction (fe_reduce_soft), so the formulas below read like the textbook ones.
//
// Memory formats (the Rust in-memory layouts): Fq2 = c0 || c1 (2 x 4 u64 Montgomery limbs); G2Affine = x || y (128 B, identity all zero);
// G2 = x || y || z Jacobian (192 B, identity z = 0).  The formulas never use the curve constant, as in ec.hpp.
#pragma once
#include "ec.hpp"

namespace zkhip {

struct fe2 {
  fe c0, c1;
};

ZK_HD fe std_form(const fe& lazy) { return fe_reduce_soft<Fq>(fe_norm(lazy)); }   // any value < 2^261 with limbs < 2^32 -> standard form

ZK_HD fe2 f2_zero() { return {fe_zero(), fe_zero()}; }
ZK_HD fe2 f2_one() { return {fe_one<Fq>(), fe_zero()}; }
ZK_HD bool f2_is_zero_limbs(const fe2& a) { return fe_is_zero_limbs(a.c0) && fe_is_zero_limbs(a.c1); }

ZK_HD fe2 f2_add(const fe2& a, const fe2& b) { return {std_form(fe_add(a.c0, b.c0)), std_form(fe_add(a.c1, b.c1))}; }
ZK_HD fe2 f2_dbl(const fe2& a) { return {std_form(fe_dbl(a.c0)), std_form(fe_dbl(a.c1))}; }
// a - b: a + 4p - b limb-wise (P4_S1 is borrow-proof for a subtrahend in N form below 3p)
ZK_HD fe2 f2_sub(const fe2& a, const fe2& b) { return {std_form(fe_sub_red(a.c0, b.c0, Fq::P4_S1)), std_form(fe_sub_red(a.c1, b.c1, Fq::P4_S1))}; }
ZK_HD fe2 f2_neg(const fe2& a) { return {std_form(fe_neg_red(a.c0, Fq::P4_S1)), std_form(fe_neg_red(a.c1, Fq::P4_S1))}; }

// Karatsuba: (a0 b0 - a1 b1) + (( a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u.  Operands in standard form: the sums are < 4p + 2^234 with limbs
// < 2^30, so every product is within fe_mul's limits and comes out N-form < 2p.
ZK_HD fe2 f2_mul(const fe2& a, const fe2& b) {
  const fe t0 = fe_mul<Fq>(a.c0, b.c0), t1 = fe_mul<Fq>(a.c1, b.c1);
  const fe s = fe_mul<Fq>(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
  fe2 r;
  r.c0 = std_form(fe_sub_red(t0, t1, Fq::P4_S1));                               // t0 - t1 + 4p
  r.c1 = std_form(fe_sub_red(s, fe_norm(fe_add(t0, t1)), Fq::P6_S1));           // s - (t0 + t1) + 6p; t0 + t1 < 4p
  return r;
}

// (a0 + a1)(a0 - a1) + 2 a0 a1 u
ZK_HD fe2 f2_sqr(const fe2& a) {
  const fe d = fe_norm(fe_sub_red(a.c0, a.c1, Fq::P4_S1));                      // a0 - a1 + 4p < 7p, N form
  const fe s = fe_add(a.c0, a.c1);
  fe2 r;
  r.c0 = std_form(fe_mul<Fq>(s, d));
  r.c1 = std_form(fe_dbl(fe_mul<Fq>(a.c0, a.c1)));
  return r;
}

// is the value 0 mod p (both components)?  One multiplication by one brings a standard-form value below 2p, where zero is 0 or p.
ZK_HD bool f2_is_zero(const fe2& a) {
  const fe one = fe_one<Fq>();
  return fe_mulout_is_zero<Fq>(fe_mul<Fq>(one, a.c0)) && fe_mulout_is_zero<Fq>(fe_mul<Fq>(one, a.c1));
}

// ---- lazy Fq2 arithmetic for the bucket accumulation (round 5) -------------------------------------------------------------------------
// The standard-form layer above ends every Fq2 operation with a ~77-instruction reduction per component (a Karatsuba product spends 385
// instructions on five of them -- more than one of its three multiplications), which put the G2 mixed addition at ~10.8 k instructions,
// 5 x G1's, where the multiplication count says 2.8 x.  Here a product is two FUSED pairs, each under ONE Montgomery reduction
// (fe_mul_add, fp29.hpp): c0 = a0 b0 + a1 (K p - b1), c1 = a0 b1 + a1 b0 -- 486 multiply-adds like Karatsuba's three products, but no
// pre-additions, no post-subtractions and no reductions in between -- and additions / subtractions stay lazy with the bounds written at
// every call site, as in ec.hpp.  `red` = K p in borrow-proof form with b.c1 < (K - 1) p, N form.
// fe_mul_add's column bound: 9 (2^29 * 2^29 + 2^29 * 2^30.1) + 9 * 2^58 < 2^63.3.
template <bool CHAIN>
ZK_HD fe2 f2_mul_lazy(const fe2& a, const fe2& b, const uint32_t (&red)[NL]) {
  fe2 r;
  r.c0 = fe_mul_add<Fq, CHAIN>(a.c0, b.c0, a.c1, fe_neg_red(b.c1, red));
  r.c1 = fe_mul_add<Fq, CHAIN>(a.c0, b.c1, a.c1, b.c0);
  return r;
}
// (a0 + a1)(a0 - a1 + K p) + 2 a0 a1 u; `red` = K p with a.c1 < (K - 1) p.  Operands N form; the sum and the doubled factor have limbs < 2^30.
template <bool CHAIN>
ZK_HD fe2 f2_sqr_lazy(const fe2& a, const uint32_t (&red)[NL]) {
  const fe d = fe_norm(fe_sub_red(a.c0, a.c1, red));
  fe2 r;
  r.c0 = fe_mul<Fq, CHAIN>(fe_add(a.c0, a.c1), d);
  r.c1 = fe_mul<Fq, CHAIN>(fe_dbl(a.c0), a.c1);
  return r;
}

// limb i of 2p in N form (limbs 0..7 < 2^29, limb 8 the rest)
template <class P> ZK_HD constexpr uint32_t two_p_limb(int i) {
  uint32_t carry = 0, out = 0;
  for (int j = 0; j <= i; j++) {
    const uint32_t t = 2u * P::P[j] + carry;
    out = j < NL - 1 ? (t & LMASK) : t;
    carry = j < NL - 1 ? (t >> LB) : 0;
  }
  return out;
}
// N-form value below 3p: is it 0 mod p?  (0, p or 2p: the N form of an integer is unique)
template <class P>
ZK_HD bool fe_is_zero_lt3p(const fe& a) {
  uint32_t o = 0, e = 0, e2 = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    o |= a.l[i];
    e |= a.l[i] ^ P::P[i];
    e2 |= a.l[i] ^ two_p_limb<P>(i);
  }
  return (o == 0) | (e == 0) | (e2 == 0);
}

// ---- points ------------------------------------------------------------------------------------------------------------
struct xyzz2 {
  fe2 X, Y, ZZ, ZZZ;        // x = X / ZZ, y = Y / ZZZ; identity <=> ZZ has all limbs zero
};

ZK_HD xyzz2 xyzz2_identity() { return {f2_zero(), f2_zero(), f2_zero(), f2_zero()}; }
ZK_HD bool xyzz2_is_identity(const xyzz2& a) { return f2_is_zero_limbs(a.ZZ); }

// 2A (dbl-2008-s-1, a = 0)
ZK_HD xyzz2 xyzz2_dbl(const xyzz2& A) {
  if (xyzz2_is_identity(A)) return A;
  const fe2 U = f2_dbl(A.Y), V = f2_sqr(U), W = f2_mul(U, V), S = f2_mul(A.X, V);
  const fe2 XX = f2_sqr(A.X), M = f2_add(f2_dbl(XX), XX);
  xyzz2 r;
  r.X = f2_sub(f2_sqr(M), f2_dbl(S));
  r.Y = f2_sub(f2_mul(M, f2_sub(S, r.X)), f2_mul(W, A.Y));
  r.ZZ = f2_mul(V, A.ZZ);
  r.ZZZ = f2_mul(W, A.ZZZ);
  return r;
}

// A + B (add-2008-s), with the doubling / opposite-point cases
ZK_HD xyzz2 xyzz2_add(const xyzz2& A, const xyzz2& B) {
  if (xyzz2_is_identity(A)) return B;
  if (xyzz2_is_identity(B)) return A;
  const fe2 U1 = f2_mul(A.X, B.ZZ), U2 = f2_mul(B.X, A.ZZ), S1 = f2_mul(A.Y, B.ZZZ), S2 = f2_mul(B.Y, A.ZZZ);
  const fe2 P = f2_sub(U2, U1), R = f2_sub(S2, S1);
  if (f2_is_zero(P)) {
    if (f2_is_zero(R)) return xyzz2_dbl(A);
    return xyzz2_identity();
  }
  const fe2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(U1, PP);
  xyzz2 r;
  r.X = f2_sub(f2_sub(f2_sqr(R), PPP), f2_dbl(Q));
  r.Y = f2_sub(f2_mul(R, f2_sub(Q, r.X)), f2_mul(S1, PPP));
  r.ZZ = f2_mul(f2_mul(A.ZZ, B.ZZ), PP);
  r.ZZZ = f2_mul(f2_mul(A.ZZZ, B.ZZZ), PPP);
  return r;
}

// acc += (x2, y2) affine, not the identity (madd-2008-s); coordinates in standard form
ZK_HD void xyzz2_madd(xyzz2& acc, const fe2& x2, const fe2& y2) {
  if (xyzz2_is_identity(acc)) {
    acc.X = x2; acc.Y = y2; acc.ZZ = f2_one(); acc.ZZZ = f2_one();
    return;
  }
  const fe2 U2 = f2_mul(x2, acc.ZZ), S2 = f2_mul(y2, acc.ZZZ);
  const fe2 P = f2_sub(U2, acc.X), R = f2_sub(S2, acc.Y);
  if (f2_is_zero(P)) {
    if (f2_is_zero(R)) {
      xyzz2 t;
      t.X = x2; t.Y = y2; t.ZZ = f2_one(); t.ZZZ = f2_one();
      acc = xyzz2_dbl(t);
    } else {
      acc = xyzz2_identity();
    }
    return;
  }
  const fe2 PP = f2_sqr(P), PPP = f2_mul(P, PP), Q = f2_mul(acc.X, PP);
  const fe2 X3 = f2_sub(f2_sub(f2_sqr(R), PPP), f2_dbl(Q));
  acc.Y = f2_sub(f2_mul(R, f2_sub(Q, X3)), f2_mul(acc.Y, PPP));
  acc.X = X3;
  acc.ZZ = f2_mul(acc.ZZ, PP);
  acc.ZZZ = f2_mul(acc.ZZZ, PPP);
}

// acc += +-(x2, y2), the mixed addition of the bucket accumulation with lazy Fq2 arithmetic.  x2, y2: the lazily unpacked external
// coordinates (N-limbed, value < 32p: fe_from_ext_lazy), negated when `neg`; the point is not the identity.
// Accumulator invariant (all components N form): X < 2p + 2^233, Y < 5p, ZZ < 2p, ZZZ < 2p; identity <=> ZZ all-zero limbs.
// 8 lazy products + 2 lazy squares = 20 reductions (standard form: 28 multiplications + ~45 reductions), ~6.3 k instructions.
template <bool CHAIN>
ZK_HD void xyzz2_madd_lazy(xyzz2& acc, const fe2& x2, const fe2& y2, bool neg) {
  if (xyzz2_is_identity(acc)) {
    acc.X = {fe_reduce_soft<Fq>(x2.c0), fe_reduce_soft<Fq>(x2.c1)};                     // < 2p + 2^233
    acc.Y = neg ? fe2{fe_reduce_soft<Fq>(fe_norm(fe_neg_red(y2.c0, Fq::P64_S1))), fe_reduce_soft<Fq>(fe_norm(fe_neg_red(y2.c1, Fq::P64_S1)))}
                : fe2{fe_reduce_soft<Fq>(y2.c0), fe_reduce_soft<Fq>(y2.c1)};
    acc.ZZ = f2_one(); acc.ZZZ = f2_one();
    return;
  }
  fe2 U2 = f2_mul_lazy<CHAIN>(acc.ZZ, x2, Fq::P64_S1);      // c0 < (2*32 + 2*64)/169.3 + 1 = 2.14p, c1 < (2*32*2)/169.3 + 1 = 1.76p
  fe2 S2 = f2_mul_lazy<CHAIN>(acc.ZZZ, y2, Fq::P64_S1);     // same
  if (neg) S2 = {fe_neg_red(S2.c0, Fq::P4_S1), fe_neg_red(S2.c1, Fq::P4_S1)};          // 4p - S2: < 4p, limbs < 2^30.1
  const fe2 P = {fe_norm(fe_sub_red(U2.c0, acc.X.c0, Fq::P4_S1)), fe_norm(fe_sub_red(U2.c1, acc.X.c1, Fq::P4_S1))};   // X < 3p; P < 6.14p
  const fe2 R = {fe_norm(fe_sub_red(S2.c0, acc.Y.c0, Fq::P6_S1)), fe_norm(fe_sub_red(S2.c1, acc.Y.c1, Fq::P6_S1))};   // Y < 5p; R < 10p
  const fe2 PP = f2_sqr_lazy<CHAIN>(P, Fq::P8_S1);          // c0 < 12.28 * 14.14 / 169.3 + 1 = 2.03p, c1 < 2 * 6.14^2 / 169.3 + 1 = 1.45p
  if (fe_is_zero_lt3p<Fq>(PP.c0) && fe_mulout_is_zero<Fq>(PP.c1)) {                    // P^2 = 0 <=> P = 0: same x (rare)
    const fe2 RR0 = f2_sqr_lazy<false>(R, Fq::P12_S1);      // < 3.6p, 2.2p
    if (fe_is_zero_lt3p<Fq>(fe_reduce_soft<Fq>(RR0.c0)) && fe_is_zero_lt3p<Fq>(RR0.c1)) {
      xyzz2 t;                                               // the same point: 2 (x2, +-y2) through the standard-form doubling
      const fe one = fe_one<Fq>();
      t.X = {fe_mul<Fq>(one, x2.c0), fe_mul<Fq>(one, x2.c1)};
      t.Y = {fe_mul<Fq>(one, y2.c0), fe_mul<Fq>(one, y2.c1)};
      if (neg) t.Y = f2_neg(t.Y);
      t.ZZ = f2_one(); t.ZZZ = f2_one();
      acc = xyzz2_dbl(t);
    } else {
      acc = xyzz2_identity();
    }
    return;
  }
  const fe2 PPP = f2_mul_lazy<CHAIN>(P, PP, Fq::P4_S1);     // < (6.14*2.03 + 6.14*4)/169.3 + 1 = 1.22p
  const fe2 Q = f2_mul_lazy<CHAIN>(acc.X, PP, Fq::P4_S1);   // < (2.01*2.03 + 2.01*4)/169.3 + 1 = 1.08p
  const fe2 RR = f2_sqr_lazy<CHAIN>(R, Fq::P12_S1);         // c0 < 20 * 22 / 169.3 + 1 = 3.6p, c1 < 2.2p
  // X3 = RR - PPP - 2Q: subtrahend < 1.22 + 2.16 = 3.4p with limbs < 3 * 2^29 -> 7p in S3 form; < 10.6p -> reduced to < 2p + 2^233 (X feeds the
  // next addition's P = U2 - X1, whose square's first factor pair grows with 4 x the bound: without this reduction the bounds do not close)
  const fe2 X3 = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c0, fe_add(PPP.c0, fe_dbl(Q.c0)), Fq::P7_S3))),
                  fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c1, fe_add(PPP.c1, fe_dbl(Q.c1)), Fq::P7_S3)))};
  const fe2 T = {fe_norm(fe_sub_red(Q.c0, X3.c0, Fq::P4_S1)), fe_norm(fe_sub_red(Q.c1, X3.c1, Fq::P4_S1))};           // < 1.08 + 4 = 5.08p, N form
  const fe2 A = f2_mul_lazy<CHAIN>(R, T, Fq::P8_S1);        // < (10*5.08 + 10*8)/169.3 + 1 = 1.78p
  const fe2 B = f2_mul_lazy<CHAIN>(acc.Y, PPP, Fq::P4_S1);  // < (5*1.22 + 5*4)/169.3 + 1 = 1.16p
  acc.Y = {fe_norm(fe_sub_red(A.c0, B.c0, Fq::P3_S1)), fe_norm(fe_sub_red(A.c1, B.c1, Fq::P3_S1))};                   // < 1.78 + 3 = 4.78p
  acc.X = X3;
  acc.ZZ = f2_mul_lazy<CHAIN>(acc.ZZ, PP, Fq::P4_S1);       // < (2*2.03 + 2*4)/169.3 + 1 = 1.08p
  acc.ZZZ = f2_mul_lazy<CHAIN>(acc.ZZZ, PPP, Fq::P4_S1);    // < 1.07p
}

// A + B (add-2008-s) with lazy Fq2 arithmetic: inputs and result in standard form (every component N form below 2p + 2^233), the
// doubling / opposite-point cases through the standard-form code.  12 lazy products + 2 lazy squares + 4 soft reductions, ~8.7 k
// instructions against ~15 k for xyzz2_add.
// 2A (dbl-2008-s-1, a = 0) with lazy Fq2 arithmetic; A in standard form through a component provider (see xyzz2_add_lazy_core), not the
// identity; result in standard form.  6 lazy products + 3 lazy squares + 4 soft reductions.
template <bool CHAIN, class GA>
ZK_HD xyzz2 xyzz2_dbl_lazy_core(GA a) {
  const fe2 AY = a(1), AX = a(0);
  const fe2 U = {fe_norm(fe_dbl(AY.c0)), fe_norm(fe_dbl(AY.c1))};          // < 4.02p
  const fe2 V = f2_sqr_lazy<CHAIN>(U, Fq::P6_S1);                            // c0 < 8.04 * 10.02 / 169.3 + 1 = 1.48p
  const fe2 W = f2_mul_lazy<CHAIN>(U, V, Fq::P4_S1);                         // < (4.02*1.48 + 4.02*4)/169.3 + 1 = 1.13p
  const fe2 S = f2_mul_lazy<CHAIN>(AX, V, Fq::P4_S1);                        // < 1.07p
  const fe2 XX = f2_sqr_lazy<CHAIN>(AX, Fq::P4_S1);                          // c0 < 4.02 * 6.01 / 169.3 + 1 = 1.14p
  const fe2 M = {fe_norm(fe_add(XX.c0, fe_dbl(XX.c0))), fe_norm(fe_add(XX.c1, fe_dbl(XX.c1)))};          // 3 x^2 < 3.42p
  const fe2 MM = f2_sqr_lazy<CHAIN>(M, Fq::P6_S1);                           // c0 < 6.84 * 9.42 / 169.3 + 1 = 1.38p
  xyzz2 r;
  r.X = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(MM.c0, fe_dbl(S.c0), Fq::P7_S3))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(MM.c1, fe_dbl(S.c1), Fq::P7_S3)))};   // 2S: limbs < 2^30
  const fe2 T = {fe_norm(fe_sub_red(S.c0, r.X.c0, Fq::P4_S1)), fe_norm(fe_sub_red(S.c1, r.X.c1, Fq::P4_S1))};      // < 5.07p, N form
  const fe2 Am = f2_mul_lazy<CHAIN>(M, T, Fq::P8_S1);                        // < (3.42*5.07 + 3.42*8)/169.3 + 1 = 1.26p
  const fe2 Bm = f2_mul_lazy<CHAIN>(W, AY, Fq::P4_S1);                       // < 1.04p
  r.Y = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c0, Bm.c0, Fq::P3_S1))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c1, Bm.c1, Fq::P3_S1)))};
  r.ZZ = f2_mul_lazy<CHAIN>(V, a(2), Fq::P4_S1);
  r.ZZZ = f2_mul_lazy<CHAIN>(W, a(3), Fq::P4_S1);
  return r;
}

// The operands come through component providers (a(0..3) = X, Y, ZZ, ZZZ), so that a kernel adding two points that live in memory loads each
// coordinate when the formula first needs it (with both 72-register operands loaded up front, and the rare equal-x case falling into the
// standard-form code, the kernels needed 256 VGPRs + 196 AGPRs + scratch: one wave per SIMD).  full_a / full_b supply a whole operand when
// the other one is the identity.
template <bool CHAIN, class GA, class GB, class FULLA, class FULLB>
ZK_HD xyzz2 xyzz2_add_lazy_core(GA a, GB b, FULLA full_a, FULLB full_b) {
  const fe2 AZZ = a(2), BZZ = b(2);
  if (f2_is_zero_limbs(AZZ)) return full_b();
  if (f2_is_zero_limbs(BZZ)) return full_a();
  const fe2 U1 = f2_mul_lazy<CHAIN>(a(0), BZZ, Fq::P4_S1), U2 = f2_mul_lazy<CHAIN>(b(0), AZZ, Fq::P4_S1);      // < (2.01^2 + 2.01*4)/169.3 + 1 = 1.08p
  const fe2 ZZ12 = f2_mul_lazy<CHAIN>(AZZ, BZZ, Fq::P4_S1);
  const fe2 P = {fe_norm(fe_sub_red(U2.c0, U1.c0, Fq::P3_S1)), fe_norm(fe_sub_red(U2.c1, U1.c1, Fq::P3_S1))};  // < 4.08p
  const fe2 PP = f2_sqr_lazy<CHAIN>(P, Fq::P6_S1);          // c0 < 8.16 * 10.08 / 169.3 + 1 = 1.49p, c1 < 1.2p
  const fe2 AZZZ = a(3), BZZZ = b(3);
  const fe2 S1 = f2_mul_lazy<CHAIN>(a(1), BZZZ, Fq::P4_S1), S2 = f2_mul_lazy<CHAIN>(b(1), AZZZ, Fq::P4_S1);
  const fe2 R = {fe_norm(fe_sub_red(S2.c0, S1.c0, Fq::P3_S1)), fe_norm(fe_sub_red(S2.c1, S1.c1, Fq::P3_S1))};
  const fe2 RR = f2_sqr_lazy<CHAIN>(R, Fq::P6_S1);          // < 1.49p
  if (fe_mulout_is_zero<Fq>(PP.c0) && fe_mulout_is_zero<Fq>(PP.c1)) {       // P^2 = 0 <=> P = 0: same x (rare)
    if (fe_mulout_is_zero<Fq>(RR.c0) && fe_mulout_is_zero<Fq>(RR.c1)) return xyzz2_dbl_lazy_core<CHAIN>(a);     // the same point
    return xyzz2_identity();                                                  // opposite points
  }
  xyzz2 r;
  r.ZZ = f2_mul_lazy<CHAIN>(ZZ12, PP, Fq::P4_S1);
  const fe2 PPP = f2_mul_lazy<CHAIN>(P, PP, Fq::P4_S1);     // < (4.08*1.49 + 4.08*4)/169.3 + 1 = 1.14p
  const fe2 Q = f2_mul_lazy<CHAIN>(U1, PP, Fq::P4_S1);      // < 1.04p
  r.ZZZ = f2_mul_lazy<CHAIN>(f2_mul_lazy<CHAIN>(AZZZ, BZZZ, Fq::P4_S1), PPP, Fq::P4_S1);
  r.X = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c0, fe_add(PPP.c0, fe_dbl(Q.c0)), Fq::P7_S3))),                // subtrahend < 3.3p, limbs < 3 * 2^29; < 8.5p -> standard form
         fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c1, fe_add(PPP.c1, fe_dbl(Q.c1)), Fq::P7_S3)))};
  const fe2 T = {fe_norm(fe_sub_red(Q.c0, r.X.c0, Fq::P4_S1)), fe_norm(fe_sub_red(Q.c1, r.X.c1, Fq::P4_S1))};    // < 5.04p, N form
  const fe2 Am = f2_mul_lazy<CHAIN>(R, T, Fq::P8_S1);       // < (4.08*5.04 + 4.08*8)/169.3 + 1 = 1.32p
  const fe2 Bm = f2_mul_lazy<CHAIN>(S1, PPP, Fq::P4_S1);    // < 1.04p
  r.Y = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c0, Bm.c0, Fq::P3_S1))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c1, Bm.c1, Fq::P3_S1)))};   // < 4.32p -> standard form
  return r;
}

template <bool CHAIN>
ZK_HD xyzz2 xyzz2_add_lazy(const xyzz2& A, const xyzz2& B) {
  auto ga = [&](int c) -> const fe2& { return c == 0 ? A.X : (c == 1 ? A.Y : (c == 2 ? A.ZZ : A.ZZZ)); };
  auto gb = [&](int c) -> const fe2& { return c == 0 ? B.X : (c == 1 ? B.Y : (c == 2 ? B.ZZ : B.ZZZ)); };
  return xyzz2_add_lazy_core<CHAIN>(ga, gb, [&]() -> xyzz2 { return A; }, [&]() -> xyzz2 { return B; });
}

// accumulator of xyzz2_madd_lazy -> standard form (every component N form below 2p + 2^233), what the other kernels' formulas expect
ZK_HD xyzz2 xyzz2_to_std(const xyzz2& a) {
  xyzz2 r = a;
  r.Y = {fe_reduce_soft<Fq>(a.Y.c0), fe_reduce_soft<Fq>(a.Y.c1)};
  return r;
}

#if defined(__HIPCC__)
// external Montgomery-256 words -> standard-form internal value (x * 2^261 mod p): one multiplication by one
ZK_D fe fq_from_ext(const uint32_t* w8) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = w8[i];
  return fe_mul<Fq>(fe_one<Fq>(), fe_from_ext_lazy(w));
}
ZK_D void fq_to_ext(const fe& a, uint32_t* out8) {
  uint32_t w[8];
  fe_to_ext<Fq>(a, w);
#pragma unroll
  for (int i = 0; i < 8; i++) out8[i] = w[i];
}

struct affine2_words { uint32_t w[32]; };   // x.c0 | x.c1 | y.c0 | y.c1

ZK_D affine2_words load_affine2(const uint32_t* base, size_t idx) {
  affine2_words a;
  const uint4* q = reinterpret_cast<const uint4*>(base + idx * 32);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint4 v = q[i];
    a.w[4 * i] = v.x; a.w[4 * i + 1] = v.y; a.w[4 * i + 2] = v.z; a.w[4 * i + 3] = v.w;
  }
  return a;
}
ZK_D bool affine2_is_identity(const affine2_words& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) o |= a.w[i];
  return o == 0;
}
ZK_D void affine2_coords(const affine2_words& a, bool negate, fe2& x, fe2& y) {
  x.c0 = fq_from_ext(a.w); x.c1 = fq_from_ext(a.w + 8);
  y.c0 = fq_from_ext(a.w + 16); y.c1 = fq_from_ext(a.w + 24);
  if (negate) y = f2_neg(y);
}

// XYZZ work format in global memory: 8 x 9 limbs (288 B)
ZK_D xyzz2 load_xyzz2(const uint32_t* base, size_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(base + idx * 72);
  uint32_t w[72];
#pragma unroll
  for (int i = 0; i < 18; i++) {
    const uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  xyzz2 r;
  fe* f[8] = {&r.X.c0, &r.X.c1, &r.Y.c0, &r.Y.c1, &r.ZZ.c0, &r.ZZ.c1, &r.ZZZ.c0, &r.ZZZ.c1};
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int i = 0; i < 9; i++) f[k]->l[i] = w[9 * k + i];
  return r;
}
// one coordinate (18 words) of a work point in memory: comp 0..3 = X, Y, ZZ, ZZZ (72 comp bytes into the record: 8-byte aligned loads)
ZK_D fe2 load_xyzz2_coord(const uint32_t* base, size_t idx, int comp) {
  const uint2* q = reinterpret_cast<const uint2*>(base + idx * 72 + comp * 18);
  uint32_t w[18];
#pragma unroll
  for (int i = 0; i < 9; i++) { const uint2 v = q[i]; w[2 * i] = v.x; w[2 * i + 1] = v.y; }
  fe2 r;
#pragma unroll
  for (int i = 0; i < 9; i++) { r.c0.l[i] = w[i]; r.c1.l[i] = w[9 + i]; }
  return r;
}
// a[ia] + b[ib], both operands in memory, coordinates loaded on demand
template <bool CHAIN>
ZK_D xyzz2 xyzz2_add_lazy_mem(const uint32_t* a, size_t ia, const uint32_t* b, size_t ib) {
  return xyzz2_add_lazy_core<CHAIN>([&](int c) { return load_xyzz2_coord(a, ia, c); }, [&](int c) { return load_xyzz2_coord(b, ib, c); },
                                    [&]() { return load_xyzz2(a, ia); }, [&]() { return load_xyzz2(b, ib); });
}
// A + b[ib], the second operand in memory
template <bool CHAIN>
ZK_D xyzz2 xyzz2_add_lazy_regmem(const xyzz2& A, const uint32_t* b, size_t ib) {
  auto ga = [&](int c) -> const fe2& { return c == 0 ? A.X : (c == 1 ? A.Y : (c == 2 ? A.ZZ : A.ZZZ)); };
  return xyzz2_add_lazy_core<CHAIN>(ga, [&](int c) { return load_xyzz2_coord(b, ib, c); }, [&]() -> xyzz2 { return A; }, [&]() { return load_xyzz2(b, ib); });
}
ZK_D void store_xyzz2(uint32_t* base, size_t idx, const xyzz2& a) {
  uint32_t w[72];
  const fe* f[8] = {&a.X.c0, &a.X.c1, &a.Y.c0, &a.Y.c1, &a.ZZ.c0, &a.ZZ.c1, &a.ZZZ.c0, &a.ZZZ.c1};
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int i = 0; i < 9; i++) w[9 * k + i] = f[k]->l[i];
  uint4* q = reinterpret_cast<uint4*>(base + idx * 72);
#pragma unroll
  for (int i = 0; i < 18; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// XYZZ -> G2 Jacobian memory (x || y || z, 48 u32 words): (X ZZ^2, Y ZZZ^2, ZZZ); identity = (0, 1, 0) like halo2curves `G2::identity()`
ZK_D void store_jacobian2(const xyzz2& a, uint32_t* out) {
  if (xyzz2_is_identity(a)) {
#pragma unroll
    for (int i = 0; i < 48; i++) out[i] = 0;
    fq_to_ext(fe_one<Fq>(), out + 16);
    return;
  }
  const fe2 x = f2_mul(a.X, f2_sqr(a.ZZ)), y = f2_mul(a.Y, f2_sqr(a.ZZZ));
  fq_to_ext(x.c0, out); fq_to_ext(x.c1, out + 8);
  fq_to_ext(y.c0, out + 16); fq_to_ext(y.c1, out + 24);
  fq_to_ext(a.ZZZ.c0, out + 32); fq_to_ext(a.ZZZ.c1, out + 40);
}
#endif

}  // namespace zkhip
