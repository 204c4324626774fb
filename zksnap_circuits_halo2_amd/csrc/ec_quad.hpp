// Quad-cooperative XYZZ arithmetic for the latency-bound tail of the MSM (bucket-sum pyramid, weighted window sum, window
// fold, multi-GPU partial fold): after the bucket accumulation only a few thousand point operations remain, each a chain of
// 9-14 dependent field multiplications executed by one lane at ~10 cycles per instruction.  Here the 4 lanes of a quad
// (lanes 4k .. 4k+3) share one point operation: every lane holds the same inputs, each formula stage's independent products
// go one per lane, and the results are exchanged with DPP quad_perm broadcasts (one VALU move per limb, no LDS):
//     add-2008-s   12M + 2S  ->  4 stages        dbl-2008-s-1   6M + 3S  ->  3 stages
// Same formulas, same operand bounds and therefore the same limb values as ec.hpp (a Montgomery product is a function of
// its operands); rare cases (an identity operand, equal or opposite points) take the single-lane code redundantly in all
// four lanes.  All four lanes of a quad must be active and hold identical arguments; all return the same result.
#pragma once
#include "ec.hpp"

namespace zkhip {

// Keep a value opaque to the optimiser.  Used on the DPP broadcasts: without it hipcc (ROCm 7.2) folds the v_mov_b32_dpp into its
// consumer, and with the broadcast value as the subtrahend of `a + K p - b` that fold produced wrong limbs (found by the parity
// test: only Y3 = R T - PPP S1 was off, while the same stage written out in a test kernel was right).  The barrier is applied to
// a by-value scalar so that the limb arrays still become registers.
ZK_D uint32_t quad_opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int J>
ZK_D fe quad_bcast(const fe& m) {   // every lane of the quad receives lane J's value
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = quad_opaque((uint32_t)__builtin_amdgcn_update_dpp(0, (int)m.l[i], J * 0x55, 0xf, 0xf, false));
  return r;
}

ZK_D fe quad_pick(uint32_t q, const fe& a0, const fe& a1, const fe& a2, const fe& a3) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    // read the four candidates first: a ternary over the lvalues becomes a select of addresses, which pins the operands in scratch
    const uint32_t v0 = a0.l[i], v1 = a1.l[i], v2 = a2.l[i], v3 = a3.l[i];
    const uint32_t lo = q & 1 ? v1 : v0, hi = q & 1 ? v3 : v2;
    r.l[i] = q & 2 ? hi : lo;
  }
  return r;
}

// one stage: lane q multiplies (a_q, b_q); p0..p3 receive the four products
#define ZK_QUAD_STAGE(q, a0, b0, a1, b1, a2, b2, a3, b3, p0, p1, p2, p3)                          \
  {                                                                                               \
    const fe m_ = fe_mul<Fq>(quad_pick(q, a0, a1, a2, a3), quad_pick(q, b0, b1, b2, b3));         \
    p0 = quad_bcast<0>(m_); p1 = quad_bcast<1>(m_); p2 = quad_bcast<2>(m_); p3 = quad_bcast<3>(m_); \
  }

// A + B, general (add-2008-s); bounds as xyzz_add
ZK_D xyzz xyzz_add_quad(const xyzz& A, const xyzz& B, uint32_t q) {
  if (xyzz_is_identity(A)) return B;
  if (xyzz_is_identity(B)) return A;
  fe U1, U2, S1, S2;
  ZK_QUAD_STAGE(q, B.ZZ, A.X, A.ZZ, B.X, B.ZZZ, A.Y, A.ZZZ, B.Y, U1, U2, S1, S2);
  const fe P = fe_norm(fe_sub_red(U2, U1, Fq::P3_S1));      // < 5p
  const fe R = fe_norm(fe_sub_red(S2, S1, Fq::P3_S1));
  fe PP, RR, ZZ12, ZZZ12;
  ZK_QUAD_STAGE(q, P, P, R, R, A.ZZ, B.ZZ, A.ZZZ, B.ZZZ, PP, RR, ZZ12, ZZZ12);
  fe PPP, Q, ZZ3, dup;
  ZK_QUAD_STAGE(q, PP, P, PP, U1, ZZ12, PP, PP, P, PPP, Q, ZZ3, dup);
  xyzz r;
  r.X = fe_norm(fe_sub_red(RR, fe_add(PPP, fe_dbl(Q)), Fq::P7_S3));
  const fe T = fe_sub_red(Q, r.X, Fq::P10_S1);
  fe RT, PS, ZZZ3;
  ZK_QUAD_STAGE(q, R, T, PPP, S1, ZZZ12, PPP, R, T, RT, PS, ZZZ3, dup);
  r.Y = fe_norm(fe_sub_red(RT, PS, Fq::P3_S1));
  r.ZZ = ZZ3;
  r.ZZZ = ZZZ3;
  // same x: doubling or opposite points (rare): the single-lane formulas, after the last cross-lane exchange
  if (fe_mulout_is_zero<Fq>(PP)) r = xyzz_add(A, B);
  return r;
}

// 2A, general (dbl-2008-s-1, a = 0); bounds as xyzz_dbl
ZK_D xyzz xyzz_dbl_quad(const xyzz& A, uint32_t q) {
  if (xyzz_is_identity(A)) return A;
  const fe U = fe_dbl(A.Y);                                 // limbs < 2^30
  fe V, XX, d0, d1;
  ZK_QUAD_STAGE(q, U, U, A.X, A.X, U, U, A.X, A.X, V, XX, d0, d1);
  const fe M = fe_norm(fe_add(XX, fe_dbl(XX)));             // 3 X^2
  fe W, S, MM, ZZ3;
  ZK_QUAD_STAGE(q, V, U, A.X, V, M, M, V, A.ZZ, W, S, MM, ZZ3);
  xyzz r;
  r.X = fe_norm(fe_sub_red(MM, fe_dbl(S), Fq::P7_S3));
  const fe T = fe_sub_red(S, r.X, Fq::P10_S1);
  fe MT, WY, ZZZ3;
  ZK_QUAD_STAGE(q, M, T, W, A.Y, W, A.ZZZ, M, T, MT, WY, ZZZ3, d0);
  r.Y = fe_norm(fe_sub_red(MT, WY, Fq::P3_S1));
  r.ZZ = ZZ3;
  r.ZZZ = ZZZ3;
  return r;
}

}  // namespace zkhip
