// BN254 G1 (y^2 = x^3 + 3) point arithmetic in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) over the lazy radix-2^29 field of fp29.hpp.
//
// Replaces the bucket arithmetic of halo2-axiom `multiexp_serial` [DEP] (mixed Jacobian+affine adds on
// bn256::G1; entered from /root/reference/aggregator/src/wrapper.rs:129).  XYZZ mixed addition costs
// 8M+2S with no inversion and, with the 7 spare bits of the 2^261 Montgomery radix, no conditional
// subtraction: every subtraction adds a borrow-proof multiple of p instead (constants P<K>_S<s>).
//
// Stored-point invariant (all coordinates in N form, see fp29.hpp):
//     X < 9p,  Y < 5p,  ZZ < 2p,  ZZZ < 2p;   identity <=> all limbs of ZZ are zero.
// The formulas never use the curve constant b, so they are valid for any a = 0 curve.
#pragma once
#include "fp29.hpp"

namespace zkhip {

using Fq = FqParams;

struct xyzz {
  fe X, Y, ZZ, ZZZ;
};

ZK_HD xyzz xyzz_identity() {
  xyzz r;
  r.X = fe_zero(); r.Y = fe_zero(); r.ZZ = fe_zero(); r.ZZZ = fe_zero();
  return r;
}

ZK_HD bool xyzz_is_identity(const xyzz& a) { return fe_is_zero_limbs(a.ZZ); }

// 2 * (x, y) for an affine point with reduced coordinates (N form, < 2p).  mdbl-2008-s-1.
ZK_HD xyzz xyzz_dbl_affine(const fe& x, const fe& y) {
  xyzz r;
  fe U = fe_dbl(y);                                   // limbs < 2^30, < 4p
  fe V = fe_sqr<Fq>(U);                               // < 1.1p
  fe W = fe_mul<Fq>(V, U);                            // N x 2^30
  fe S = fe_mul<Fq>(x, V);
  fe XX = fe_sqr<Fq>(x);                              // < 1.03p
  fe M = fe_norm(fe_add(XX, fe_dbl(XX)));             // 3 x^2 < 3.1p, N
  fe MM = fe_sqr<Fq>(M);
  r.X = fe_norm(fe_sub_red(MM, fe_dbl(S), Fq::P7_S3));                 // limbs(2S) < 2^30 <= 3*2^29; < 9p
  fe T = fe_sub_red(S, r.X, Fq::P10_S1);                               // limbs < 1.5*2^30, < 12p
  r.Y = fe_norm(fe_sub_red(fe_mul<Fq>(M, T), fe_mul<Fq>(W, y), Fq::P3_S1));  // < 5p
  r.ZZ = V;
  r.ZZZ = W;
  return r;
}

// 2 * A (general).  dbl-2008-s-1 with a = 0.
ZK_HD xyzz xyzz_dbl(const xyzz& A) {
  if (xyzz_is_identity(A)) return A;
  xyzz r;
  fe U = fe_dbl(A.Y);                                 // limbs < 2^30, < 10p
  fe V = fe_sqr<Fq>(U);                               // 100/169 + 1 < 1.6p
  fe W = fe_mul<Fq>(V, U);
  fe S = fe_mul<Fq>(A.X, V);                          // 9*1.6/169 + 1
  fe XX = fe_sqr<Fq>(A.X);                            // 81/169 + 1 < 1.5p
  fe M = fe_norm(fe_add(XX, fe_dbl(XX)));             // < 4.5p
  fe MM = fe_sqr<Fq>(M);
  r.X = fe_norm(fe_sub_red(MM, fe_dbl(S), Fq::P7_S3));
  fe T = fe_sub_red(S, r.X, Fq::P10_S1);
  r.Y = fe_norm(fe_sub_red(fe_mul<Fq>(M, T), fe_mul<Fq>(W, A.Y), Fq::P3_S1));
  r.ZZ = fe_mul<Fq>(V, A.ZZ);
  r.ZZZ = fe_mul<Fq>(W, A.ZZZ);
  return r;
}

// acc += (x2, y2): mixed addition madd-2008-s.  x2, y2 are N-limbed and may be unreduced up to 64p
// (the lazy external load gives < 32p, its negation < 64p).  The affine point must not be the identity.
// CHAIN selects the multiply shape of the common path (fp29.hpp fe_mac).
template <bool CHAIN = false>
ZK_HD void xyzz_madd_nz(xyzz& acc, const fe& x2, const fe& y2) {   // acc is not the identity
  fe U2 = fe_mul<Fq, CHAIN>(acc.ZZ, x2);                     // 2*64/169 + 1 < 1.8p
  fe S2 = fe_mul<Fq, CHAIN>(acc.ZZZ, y2);
  fe P = fe_norm(fe_sub_red(U2, acc.X, Fq::P10_S1));  // X1 N < 9p;  P < 12p
  fe R = fe_norm(fe_sub_red(S2, acc.Y, Fq::P6_S1));   // Y1 N < 5p;  R < 8p
  fe PP = fe_sqr<Fq, CHAIN>(P);                              // 144/169 + 1 < 1.86p
  if (fe_mulout_is_zero<Fq>(PP)) {                    // same x: doubling or inverse points (rare)
    fe RR0 = fe_sqr<Fq>(R);
    if (fe_mulout_is_zero<Fq>(RR0)) {
      fe one = fe_one<Fq>();
      acc = xyzz_dbl_affine(fe_mul<Fq>(one, x2), fe_mul<Fq>(one, y2));
    } else {
      acc = xyzz_identity();
    }
    return;
  }
  fe PPP = fe_mul<Fq, CHAIN>(PP, P);                         // < 1.14p
  fe Q = fe_mul<Fq, CHAIN>(PP, acc.X);                       // < 1.1p
  fe RR = fe_sqr<Fq, CHAIN>(R);                              // < 1.4p
  fe X3 = fe_norm(fe_sub_red(RR, fe_add(PPP, fe_dbl(Q)), Fq::P7_S3));  // subtrahend limbs < 3*2^29; X3 < 9p
  fe T = fe_sub_red(Q, X3, Fq::P10_S1);               // limbs < 1.5*2^30, < 12p
  fe Y3;
  if constexpr (CHAIN) {
    // R T + PPP (6p - Y1) under one reduction: columns 9 * (2^29 * 1.5 * 2^30 + 2^29 * 2^30) + 9 * 2^58 = 27 * 2^59 < 2^64;
    // value < ((8 * 12 + 1.14 * 6) / 169 + 1) p < 1.7p
    Y3 = fe_mul_add<Fq, true>(R, T, PPP, fe_neg_red(acc.Y, Fq::P6_S1));
  } else {
    Y3 = fe_norm(fe_sub_red(fe_mul<Fq>(R, T), fe_mul<Fq>(PPP, acc.Y), Fq::P3_S1));  // < 5p
  }
  acc.ZZ = fe_mul<Fq, CHAIN>(acc.ZZ, PP);
  acc.ZZZ = fe_mul<Fq, CHAIN>(acc.ZZZ, PPP);
  acc.X = X3;
  acc.Y = Y3;
}

template <bool CHAIN = false>
ZK_HD void xyzz_madd(xyzz& acc, const fe& x2, const fe& y2) {
  if (xyzz_is_identity(acc)) {
    fe one = fe_one<Fq>();
    acc.X = fe_mul<Fq>(one, x2);                      // reduce: < 1.4p
    acc.Y = fe_mul<Fq>(one, y2);
    acc.ZZ = one;
    acc.ZZZ = one;
    return;
  }
  xyzz_madd_nz<CHAIN>(acc, x2, y2);
}

// A + B (general).  add-2008-s.
ZK_HD xyzz xyzz_add(const xyzz& A, const xyzz& B) {
  if (xyzz_is_identity(A)) return B;
  if (xyzz_is_identity(B)) return A;
  fe U1 = fe_mul<Fq>(B.ZZ, A.X);                      // 2*9/169 + 1 < 1.11p
  fe U2 = fe_mul<Fq>(A.ZZ, B.X);
  fe S1 = fe_mul<Fq>(B.ZZZ, A.Y);
  fe S2 = fe_mul<Fq>(A.ZZZ, B.Y);
  fe P = fe_norm(fe_sub_red(U2, U1, Fq::P3_S1));      // < 5p
  fe R = fe_norm(fe_sub_red(S2, S1, Fq::P3_S1));
  fe PP = fe_sqr<Fq>(P);
  if (fe_mulout_is_zero<Fq>(PP)) {
    fe RR0 = fe_sqr<Fq>(R);
    if (fe_mulout_is_zero<Fq>(RR0)) return xyzz_dbl(A);
    return xyzz_identity();
  }
  fe PPP = fe_mul<Fq>(PP, P);
  fe Q = fe_mul<Fq>(PP, U1);
  fe RR = fe_sqr<Fq>(R);
  xyzz r;
  r.X = fe_norm(fe_sub_red(RR, fe_add(PPP, fe_dbl(Q)), Fq::P7_S3));
  fe T = fe_sub_red(Q, r.X, Fq::P10_S1);
  r.Y = fe_norm(fe_sub_red(fe_mul<Fq>(R, T), fe_mul<Fq>(PPP, S1), Fq::P3_S1));
  r.ZZ = fe_mul<Fq>(fe_mul<Fq>(A.ZZ, B.ZZ), PP);
  r.ZZZ = fe_mul<Fq>(fe_mul<Fq>(A.ZZZ, B.ZZZ), PPP);
  return r;
}

// ---- external formats ---------------------------------------------------------------------------
// G1Affine memory (x||y, 8 x u64 Montgomery-256 limbs, (0,0) = identity) as 16 u32 words.
#if defined(__HIPCC__)
struct affine_words {
  uint32_t x[8], y[8];
};

ZK_D affine_words load_affine(const uint32_t* base, size_t idx) {
  affine_words a;
  const uint4* q = reinterpret_cast<const uint4*>(base + idx * 16);
  uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
  a.x[0] = v0.x; a.x[1] = v0.y; a.x[2] = v0.z; a.x[3] = v0.w;
  a.x[4] = v1.x; a.x[5] = v1.y; a.x[6] = v1.z; a.x[7] = v1.w;
  a.y[0] = v2.x; a.y[1] = v2.y; a.y[2] = v2.z; a.y[3] = v2.w;
  a.y[4] = v3.x; a.y[5] = v3.y; a.y[6] = v3.z; a.y[7] = v3.w;
  return a;
}

ZK_D bool affine_is_identity(const affine_words& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.x[i] | a.y[i];
  return o == 0;
}

// XYZZ -> G1 Jacobian memory (x||y||z, 24 u32 words, canonical Montgomery-256; identity = (0, 1, 0)
// like halo2curves `G1::identity()` [DEP]).  (X, Y, ZZ, ZZZ) -> (X ZZ^2, Y ZZZ^2, ZZZ).
ZK_D void store_jacobian(const xyzz& a, uint32_t* out) {
  uint32_t w[8];
  if (xyzz_is_identity(a)) {
#pragma unroll
    for (int i = 0; i < 24; i++) out[i] = 0;
    fe one = fe_one<Fq>();
    fe_to_ext<Fq>(one, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[8 + i] = w[i];
    return;
  }
  fe zz2 = fe_sqr<Fq>(a.ZZ);
  fe zzz2 = fe_sqr<Fq>(a.ZZZ);
  fe_to_ext<Fq>(fe_mul<Fq>(zz2, a.X), w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = w[i];
  fe_to_ext<Fq>(fe_mul<Fq>(zzz2, a.Y), w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[8 + i] = w[i];
  fe_to_ext<Fq>(a.ZZZ, w);
#pragma unroll
  for (int i = 0; i < 8; i++) out[16 + i] = w[i];
}

// G1 Jacobian memory -> XYZZ (ZZ = Z^2, ZZZ = Z^3), coordinates reduced.
ZK_D xyzz load_jacobian(const uint32_t* in) {
  uint32_t w[8];
  fe one = fe_one<Fq>();
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = in[16 + i];
  fe Z = fe_mul<Fq>(one, fe_from_ext_lazy(w));
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= w[i];
  if (o == 0) return xyzz_identity();
  xyzz r;
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = in[i];
  r.X = fe_mul<Fq>(one, fe_from_ext_lazy(w));
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = in[8 + i];
  r.Y = fe_mul<Fq>(one, fe_from_ext_lazy(w));
  r.ZZ = fe_sqr<Fq>(Z);
  r.ZZZ = fe_mul<Fq>(r.ZZ, Z);
  return r;
}
#endif

}  // namespace zkhip
