// Quad-cooperative Fq2 / G2 arithmetic for the latency-bound end of the G2 MSM (per-bucket combine, pyramid, window sums, the window
// fold's 240 dependent doublings): the four lanes of a quad share every Fq2 multiplication -- lane q computes one of the four
// products a_i b_j (two for a squaring), DPP broadcasts hand all four to every lane, and each lane finishes the combination itself --
// so an Fq2 product costs the latency of ONE base-field multiplication instead of Karatsuba's three in sequence.  Point formulas are
// those of ec2.hpp with these products.  Results are the same residues as ec2.hpp's, in standard form; the representatives may differ
// (schoolbook a0 b1 + a1 b0 here, Karatsuba there), which no caller depends on.
// All four lanes of a quad must be active and hold identical arguments; all return the same result.
#pragma once
#include "ec2.hpp"
#include "ec_quad.hpp"

namespace zkhip {

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u
ZK_D fe2 f2_mul_quad(const fe2& a, const fe2& b, uint32_t q) {
  const fe m = fe_mul<Fq>(quad_pick(q, a.c0, a.c1, a.c0, a.c1), quad_pick(q, b.c0, b.c1, b.c1, b.c0));
  const fe p00 = quad_bcast<0>(m), p11 = quad_bcast<1>(m), p01 = quad_bcast<2>(m), p10 = quad_bcast<3>(m);
  fe2 r;
  r.c0 = std_form(fe_sub_red(p00, p11, Fq::P4_S1));          // products are N-form < 2p
  r.c1 = std_form(fe_add(p01, p10));
  return r;
}

// (a0 + a1)(a0 - a1) + 2 a0 a1 u: two products, on lanes 0 and 2
ZK_D fe2 f2_sqr_quad(const fe2& a, uint32_t q) {
  const fe d = fe_norm(fe_sub_red(a.c0, a.c1, Fq::P4_S1));   // a0 - a1 + 4p < 7p, N form
  const fe s = fe_add(a.c0, a.c1);
  const fe m = fe_mul<Fq>(quad_pick(q, s, s, a.c0, a.c0), quad_pick(q, d, d, a.c1, a.c1));
  fe2 r;
  r.c0 = std_form(quad_bcast<0>(m));
  r.c1 = std_form(fe_dbl(quad_bcast<2>(m)));
  return r;
}

// 2A (dbl-2008-s-1, a = 0), as xyzz2_dbl
ZK_D xyzz2 xyzz2_dbl_quad(const xyzz2& A, uint32_t q) {
  if (xyzz2_is_identity(A)) return A;
  const fe2 U = f2_dbl(A.Y), V = f2_sqr_quad(U, q), W = f2_mul_quad(U, V, q), S = f2_mul_quad(A.X, V, q);
  const fe2 XX = f2_sqr_quad(A.X, q), M = f2_add(f2_dbl(XX), XX);
  xyzz2 r;
  r.X = f2_sub(f2_sqr_quad(M, q), f2_dbl(S));
  r.Y = f2_sub(f2_mul_quad(M, f2_sub(S, r.X), q), f2_mul_quad(W, A.Y, q));
  r.ZZ = f2_mul_quad(V, A.ZZ, q);
  r.ZZZ = f2_mul_quad(W, A.ZZZ, q);
  return r;
}

// A + B (add-2008-s), as xyzz2_add; equal / opposite points take the single-lane code in all four lanes (rare)
ZK_D xyzz2 xyzz2_add_quad(const xyzz2& A, const xyzz2& B, uint32_t q) {
  if (xyzz2_is_identity(A)) return B;
  if (xyzz2_is_identity(B)) return A;
  const fe2 U1 = f2_mul_quad(A.X, B.ZZ, q), U2 = f2_mul_quad(B.X, A.ZZ, q), S1 = f2_mul_quad(A.Y, B.ZZZ, q), S2 = f2_mul_quad(B.Y, A.ZZZ, q);
  const fe2 P = f2_sub(U2, U1), R = f2_sub(S2, S1);
  if (f2_is_zero(P)) return xyzz2_add(A, B);                 // (the zero test is a function of the shared inputs: uniform over the quad)
  const fe2 PP = f2_sqr_quad(P, q), PPP = f2_mul_quad(P, PP, q), Q = f2_mul_quad(U1, PP, q);
  xyzz2 r;
  r.X = f2_sub(f2_sub(f2_sqr_quad(R, q), PPP), f2_dbl(Q));
  r.Y = f2_sub(f2_mul_quad(R, f2_sub(Q, r.X), q), f2_mul_quad(S1, PPP, q));
  r.ZZ = f2_mul_quad(f2_mul_quad(A.ZZ, B.ZZ, q), PP, q);
  r.ZZZ = f2_mul_quad(f2_mul_quad(A.ZZZ, B.ZZZ, q), PPP, q);
  return r;
}

// ---- lazy quad arithmetic (round 5): the window sums and the window fold are chains of dependent quad operations, and the standard-form layer
// above ends every Fq2 operation with a 77-instruction reduction per component ON the critical path (a doubling: 9 products + 6 additions =
// ~2300 of its ~4600 chained instructions).  Here products come back with their components merely normalised -- (a0 b0 + 3p - a1 b1, a0 b1 + a1 b0):
// c0 < 5p, c1 < 4p -- additions stay lazy with the bounds written at every call site, and only X and Y of a result (which the next operation
// subtracts from) take the quotient-estimate reduction.  Invariant of a lazy quad point: X, Y in standard form (< 2p + 2^233), ZZ and ZZZ
// components N form below 5p (identity <=> ZZ all-zero limbs); xyzz2_quad_to_std brings all four back to standard form for the other kernels.
ZK_D fe2 f2_mul_quad_lazy(const fe2& a, const fe2& b, uint32_t q) {        // every product a_i b_j below 169 p^2: each < 2p
  const fe m = fe_mul<Fq>(quad_pick(q, a.c0, a.c1, a.c0, a.c1), quad_pick(q, b.c0, b.c1, b.c1, b.c0));
  const fe p00 = quad_bcast<0>(m), p11 = quad_bcast<1>(m), p01 = quad_bcast<2>(m), p10 = quad_bcast<3>(m);
  fe2 r;
  r.c0 = fe_norm(fe_sub_red(p00, p11, Fq::P3_S1));          // < 2p + 3p
  r.c1 = fe_norm(fe_add(p01, p10));                         // < 4p
  return r;
}
// (a0 + a1)(a0 - a1 + K p) + 2 a0 a1 u; red = K p with a.c1 < (K - 1) p; (2 Ba)(Ba + K) <= 169.  c0 < 2p, c1 < 4p.
ZK_D fe2 f2_sqr_quad_lazy(const fe2& a, uint32_t q, const uint32_t (&red)[NL]) {
  const fe d = fe_norm(fe_sub_red(a.c0, a.c1, red));
  const fe s = fe_add(a.c0, a.c1);                          // limbs < 2^30
  const fe m = fe_mul<Fq>(quad_pick(q, s, s, a.c0, a.c0), quad_pick(q, d, d, a.c1, a.c1));
  fe2 r;
  r.c0 = quad_bcast<0>(m);
  r.c1 = fe_norm(fe_dbl(quad_bcast<2>(m)));
  return r;
}
ZK_D fe2 f2_soft(const fe2& a) { return {fe_reduce_soft<Fq>(a.c0), fe_reduce_soft<Fq>(a.c1)}; }      // N form, < 2^261 -> standard form
ZK_D xyzz2 xyzz2_quad_to_std(const xyzz2& A) { return {f2_soft(A.X), f2_soft(A.Y), f2_soft(A.ZZ), f2_soft(A.ZZZ)}; }

// 2A (dbl-2008-s-1) on a lazy quad point; result a lazy quad point
ZK_D xyzz2 xyzz2_dbl_quad_lazy(const xyzz2& A, uint32_t q) {
  if (xyzz2_is_identity(A)) return A;
  const fe2 U = {fe_norm(fe_dbl(A.Y.c0)), fe_norm(fe_dbl(A.Y.c1))};                       // < 4.02p
  const fe2 V = f2_sqr_quad_lazy(U, q, Fq::P6_S1);                                         // 8.04 * 10.02 / 169.3 + 1: c0 < 1.48p, c1 < 2.2p
  const fe2 W = f2_mul_quad_lazy(U, V, q);                                                 // (5, 4)
  const fe2 S = f2_mul_quad_lazy(A.X, V, q);                                               // (5, 4)
  const fe2 XX = f2_sqr_quad_lazy(A.X, q, Fq::P4_S1);                                      // c0 < 1.14p, c1 < 2.1p
  const fe2 M = {fe_norm(fe_add(XX.c0, fe_dbl(XX.c0))), fe_norm(fe_add(XX.c1, fe_dbl(XX.c1)))};           // 3 x^2: (3.42, 6.3)
  const fe2 MM = f2_sqr_quad_lazy(M, q, Fq::P8_S1);                                        // 9.72 * 11.42 / 169.3 + 1: c0 < 1.66p, c1 < 2.3p
  xyzz2 r;
  // X3 = MM - 2S: 2S < (10, 8) p with limbs < 2^30 -> 12p in S3 form; < 14.3p -> standard form
  r.X = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(MM.c0, fe_dbl(S.c0), Fq::P12_S3))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(MM.c1, fe_dbl(S.c1), Fq::P12_S3)))};
  const fe2 T = {fe_norm(fe_sub_red(S.c0, r.X.c0, Fq::P4_S1)), fe_norm(fe_sub_red(S.c1, r.X.c1, Fq::P4_S1))};    // (9, 8)
  const fe2 Am = f2_mul_quad_lazy(M, T, q);                                                // 6.3 * 9 / 169.3 + 1 < 2
  const fe2 Bm = f2_mul_quad_lazy(W, A.Y, q);
  r.Y = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c0, Bm.c0, Fq::P6_S1))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c1, Bm.c1, Fq::P6_S1)))};   // Bm < 5p; < 11p -> standard form
  r.ZZ = f2_mul_quad_lazy(V, A.ZZ, q);                                                     // 2.2 * 5 / 169.3 + 1 < 2: (5, 4)
  r.ZZZ = f2_mul_quad_lazy(W, A.ZZZ, q);                                                   // 5 * 5 / 169.3 + 1 < 2: (5, 4)
  return r;
}

// A + B (add-2008-s): A a lazy quad point, B in standard form (a stored point); result a lazy quad point.  Equal / opposite points take the
// standard-form single-lane code in all four lanes (rare).
ZK_D xyzz2 xyzz2_add_quad_lazy(const xyzz2& A, const xyzz2& B, uint32_t q) {
  if (xyzz2_is_identity(A)) return B;
  if (xyzz2_is_identity(B)) return A;
  const fe2 U1 = f2_mul_quad_lazy(A.X, B.ZZ, q), U2 = f2_mul_quad_lazy(B.X, A.ZZ, q);      // (5, 4)
  const fe2 S1 = f2_mul_quad_lazy(A.Y, B.ZZZ, q), S2 = f2_mul_quad_lazy(B.Y, A.ZZZ, q);
  const fe2 P = f2_soft({fe_norm(fe_sub_red(U2.c0, U1.c0, Fq::P6_S1)), fe_norm(fe_sub_red(U2.c1, U1.c1, Fq::P6_S1))});   // < 11p -> standard form
  const fe2 R = f2_soft({fe_norm(fe_sub_red(S2.c0, S1.c0, Fq::P6_S1)), fe_norm(fe_sub_red(S2.c1, S1.c1, Fq::P6_S1))});
  if (fe_is_zero_lt3p<Fq>(P.c0) && fe_is_zero_lt3p<Fq>(P.c1)) return xyzz2_add(xyzz2_quad_to_std(A), B);     // uniform over the quad
  const fe2 PP = f2_sqr_quad_lazy(P, q, Fq::P4_S1);                                        // c0 < 1.14p, c1 < 2.1p
  const fe2 PPP = f2_mul_quad_lazy(P, PP, q), Q = f2_mul_quad_lazy(U1, PP, q);             // (5, 4)
  const fe2 RR = f2_sqr_quad_lazy(R, q, Fq::P4_S1);
  xyzz2 r;
  // X3 = RR - PPP - 2Q: subtrahend < (15, 12) p with limbs < 3 * 2^29 -> 16p in S3 form; < 18.1p -> standard form
  r.X = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c0, fe_add(PPP.c0, fe_dbl(Q.c0)), Fq::P16_S3))),
         fe_reduce_soft<Fq>(fe_norm(fe_sub_red(RR.c1, fe_add(PPP.c1, fe_dbl(Q.c1)), Fq::P16_S3)))};
  const fe2 T = {fe_norm(fe_sub_red(Q.c0, r.X.c0, Fq::P4_S1)), fe_norm(fe_sub_red(Q.c1, r.X.c1, Fq::P4_S1))};    // (9, 8)
  const fe2 Am = f2_mul_quad_lazy(R, T, q), Bm = f2_mul_quad_lazy(S1, PPP, q);             // 5 * 5 / 169.3 + 1 < 2
  r.Y = {fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c0, Bm.c0, Fq::P6_S1))), fe_reduce_soft<Fq>(fe_norm(fe_sub_red(Am.c1, Bm.c1, Fq::P6_S1)))};
  r.ZZ = f2_mul_quad_lazy(f2_mul_quad_lazy(A.ZZ, B.ZZ, q), PP, q);                         // (5 * 2.01, then 5 * 2.1) / 169.3 + 1 < 2: (5, 4)
  r.ZZZ = f2_mul_quad_lazy(f2_mul_quad_lazy(A.ZZZ, B.ZZZ, q), PPP, q);                     // 5 * 5 / 169.3 + 1 < 2
  return r;
}

}  // namespace zkhip
