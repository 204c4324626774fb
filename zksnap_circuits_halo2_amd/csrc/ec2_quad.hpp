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

}  // namespace zkhip
