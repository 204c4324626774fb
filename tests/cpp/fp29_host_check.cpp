// CPU unit test of the *device* arithmetic: csrc/fp29.hpp and csrc/ec.hpp are host-compilable, so the lazy radix-2^29
// field and the XYZZ point formulas are exercised here (built with -fsanitize=address,undefined by tests/test_host_arith.py)
// against an independent schoolbook big-integer implementation written in this file.  Besides equality it checks the
// magnitude discipline the GPU kernels rely on: every stored coordinate is in N form with the documented value bound.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ec.hpp"
#include "fe_inverse.hpp"

using namespace zkhip;
typedef unsigned __int128 u128;

// ---- tiny big-int reference: 320-bit little-endian numbers in 5 x u64 ---------------------------------------------
struct big { uint64_t w[5]; };
static big big_zero() { big r; memset(&r, 0, sizeof(r)); return r; }
static int big_cmp(const big& a, const big& b) { for (int i = 4; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1; return 0; }
static big big_sub(const big& a, const big& b) { big r; uint64_t br = 0; for (int i = 0; i < 5; i++) { u128 d = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return r; }
static big big_add(const big& a, const big& b) { big r; uint64_t c = 0; for (int i = 0; i < 5; i++) { u128 s = (u128)a.w[i] + b.w[i] + c; r.w[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } return r; }
static big big_shl1(const big& a) { big r; uint64_t c = 0; for (int i = 0; i < 5; i++) { r.w[i] = (a.w[i] << 1) | c; c = a.w[i] >> 63; } return r; }
static big big_mod(big a, const big& p) { while (big_cmp(a, p) >= 0) a = big_sub(a, p); return a; }
// a * b mod p for a, b < p < 2^255, odd p: 4 x 64-bit Montgomery (CIOS) with R = 2^256, then one more Montgomery multiply by
// R^2 to leave the Montgomery domain -- a different radix and algorithm from the code under test.
static uint64_t neg_inv64(uint64_t p0) { uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - p0 * x; return (uint64_t)0 - x; }
static big mont4(const big& a, const big& b, const big& p, uint64_t inv) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)a.w[j] * b.w[i] + t[j] + carry; t[j] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
    u128 s = (u128)t[4] + carry; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
    const uint64_t m = t[0] * inv;
    s = (u128)m * p.w[0] + t[0]; carry = (uint64_t)(s >> 64);
    for (int j = 1; j < 4; j++) { s = (u128)m * p.w[j] + t[j] + carry; t[j - 1] = (uint64_t)s; carry = (uint64_t)(s >> 64); }
    s = (u128)t[4] + carry; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
  }
  big r = big_zero();
  for (int i = 0; i < 5; i++) r.w[i] = t[i];
  return big_mod(r, p);
}
static big r2_for(const big& p) {   // 2^512 mod p by repeated doubling (once per modulus)
  big r = big_zero(); r.w[0] = 1;
  for (int i = 0; i < 512; i++) r = big_mod(big_shl1(r), p);
  return r;
}
static big mulmod(const big& a, const big& b, const big& p) {
  static big cached_p = big_zero(), cached_r2 = big_zero();
  static uint64_t cached_inv = 0;
  if (big_cmp(cached_p, p) != 0) { cached_p = p; cached_r2 = r2_for(p); cached_inv = neg_inv64(p.w[0]); }
  return mont4(mont4(a, b, p, cached_inv), cached_r2, p, cached_inv);   // (ab/R) * R^2 / R = ab
}
static big addmod(const big& a, const big& b, const big& p) { return big_mod(big_add(a, b), p); }
static big submod(const big& a, const big& b, const big& p) { return big_cmp(a, b) >= 0 ? big_sub(a, b) : big_sub(big_add(a, p), b); }
static big powmod(big a, const big& e, const big& p) { big r = big_zero(); r.w[0] = 1; for (int i = 0; i < 256; i++) { if ((e.w[i >> 6] >> (i & 63)) & 1) r = mulmod(r, a, p); a = mulmod(a, a, p); } return r; }
static big invmod(const big& a, const big& p) { big e = p; big two = big_zero(); two.w[0] = 2; e = big_sub(e, two); return powmod(a, e, p); }

template <class P> static big modulus() { big p = big_zero(); for (int i = 0; i < NL; i++) { int bit = LB * i; p.w[bit >> 6] |= (uint64_t)P::P[i] << (bit & 63); if ((bit & 63) + LB > 64 && (bit >> 6) + 1 < 5) p.w[(bit >> 6) + 1] |= (uint64_t)P::P[i] >> (64 - (bit & 63)); } return p; }
static big fe_value(const fe& a) {   // integer value of a (possibly unnormalised) limb vector
  big r = big_zero();
  for (int i = 0; i < NL; i++) { big t = big_zero(); int bit = LB * i; t.w[bit >> 6] = (uint64_t)a.l[i] << (bit & 63); if ((bit & 63) + 32 > 64) t.w[(bit >> 6) + 1] = (uint64_t)a.l[i] >> (64 - (bit & 63)); r = big_add(r, t); }
  return r;
}
static fe fe_from_big(const big& v) { fe r; for (int i = 0; i < NL; i++) { int bit = LB * i; uint64_t x = v.w[bit >> 6] >> (bit & 63); if ((bit & 63) + LB > 64) x |= v.w[(bit >> 6) + 1] << (64 - (bit & 63)); r.l[i] = (uint32_t)x & LMASK; } return r; }

static uint64_t rng_state = 0x5A4B534E41500009ULL;
static uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ULL; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static big rnd_below(const big& p) { big r; for (int i = 0; i < 4; i++) r.w[i] = rnd(); r.w[4] = 0; r.w[3] &= (1ULL << 62) - 1; return big_mod(r, p); }

static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { if (fails < 20) printf("FAIL %s (line %d)\n", msg, __LINE__); fails++; } } while (0)

template <class P> static big R261() { big r = big_zero(); r.w[4] = 1ULL << (261 - 256); return big_mod(r, modulus<P>()); }
template <class P> static big to_int(const fe& a, const big& rinv) { return mulmod(big_mod(fe_value(a), modulus<P>()), rinv, modulus<P>()); }   // Montgomery-261 -> plain
template <class P> static fe to_fe(const big& x) { return fe_from_big(mulmod(x, R261<P>(), modulus<P>())); }                                  // plain -> Montgomery-261, canonical

static bool is_N(const fe& a) { for (int i = 0; i < NL - 1; i++) if (a.l[i] > LMASK) return false; return true; }
static bool below(const fe& a, unsigned k, const big& p) { big kp = big_zero(); for (unsigned i = 0; i < k; i++) kp = big_add(kp, p); return big_cmp(fe_value(a), kp) < 0; }

template <class P> static void field_tests(const char* name) {
  const big p = modulus<P>(), rinv = invmod(R261<P>(), p);
  for (int it = 0; it < 3000; it++) {
    big x = rnd_below(p), y = rnd_below(p);
    if (it == 0) { x = big_zero(); }
    if (it == 1) { x = big_sub(p, [] { big o = big_zero(); o.w[0] = 1; return o; }()); y = x; }
    fe a = to_fe<P>(x), b = to_fe<P>(y);
    fe m = fe_mul<P>(a, b), s = fe_sqr<P>(a);
    CHECK(is_N(m) && below(m, 2, p) && big_cmp(to_int<P>(m, rinv), mulmod(x, y, p)) == 0, name);
    CHECK(is_N(s) && below(s, 2, p) && big_cmp(to_int<P>(s, rinv), mulmod(x, x, p)) == 0, name);
    // lazy operands: sums of a few elements (limbs < 2^31) times a normalised one, as the point formulas use them
    fe lazy = fe_add(fe_add(a, b), fe_add(a, a));
    CHECK(big_cmp(to_int<P>(fe_mul<P>(b, lazy), rinv), mulmod(y, addmod(addmod(x, y, p), addmod(x, x, p), p), p)) == 0, name);
    CHECK(big_cmp(to_int<P>(fe_sub_red(a, b, P::P3_S1), rinv), submod(x, y, p)) == 0, name);
    CHECK(big_cmp(to_int<P>(fe_norm(lazy), rinv), to_int<P>(lazy, rinv)) == 0 && is_N(fe_norm(lazy)), name);
    fe c = fe_canon<P>(lazy);
    CHECK(below(c, 1, p) && big_cmp(to_int<P>(c, rinv), to_int<P>(lazy, rinv)) == 0, name);
    fe soft = fe_reduce_soft<P>(fe_norm(fe_add(lazy, lazy)));
    CHECK(below(soft, 3, p) && big_cmp(to_int<P>(soft, rinv), to_int<P>(fe_add(lazy, lazy), rinv)) == 0, name);
    CHECK(fe_mulout_is_zero<P>(fe_mul<P>(a, fe_zero())), name);
    // external format round trip: words of x*2^256 <-> internal
    big r256 = big_zero(); r256.w[4] = 1; r256 = big_mod(r256, p);
    big ext = mulmod(x, r256, p);
    uint32_t w[8], w2[8];
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)(ext.w[i >> 1] >> ((i & 1) * 32));
    fe in = fe_from_ext_lazy(w);
    CHECK(big_cmp(to_int<P>(in, rinv), x) == 0, name);
    fe_to_ext<P>(in, w2);
    CHECK(memcmp(w, w2, 32) == 0, name);
  }
  // inversion by division steps (fe_inverse.hpp) against the big-integer a^(p-2): plain integers and Montgomery-261 values, edge values,
  // and lazily reduced inputs (0 -> 0)
  const big one_b = [] { big o = big_zero(); o.w[0] = 1; return o; }();
  for (int it = 0; it < 600; it++) {
    big x = rnd_below(p);
    if (it == 0) x = big_zero();
    if (it == 1) x = one_b;
    if (it == 2) x = big_sub(p, one_b);
    if (it == 3) { x = big_zero(); x.w[0] = 2; }
    if (it == 4) x = big_sub(p, big_add(one_b, one_b));
    if (it >= 5 && it < 40) { x = big_zero(); x.w[(it - 5) / 9] = 1ULL << (7 * ((it - 5) % 9)); x = big_mod(x, p); }      // single bits
    if (it >= 40 && it < 60) { x = rnd_below(p); x.w[3] = 0; x.w[2] = 0; if (it & 1) x.w[1] = 0; }                       // short values
    const big want = big_cmp(x, big_zero()) == 0 ? big_zero() : invmod(x, p);
    const fe plain = fe_inverse_plain<P>(fe_from_big(x));
    CHECK(is_N(plain) && below(plain, 1, p) && big_cmp(fe_value(plain), want) == 0, name);
    const fe a = to_fe<P>(x);
    const fe inv = fe_inverse<P>(a);
    CHECK(is_N(inv) && below(inv, 2, p) && big_cmp(to_int<P>(inv, rinv), want) == 0, name);
    fe k; for (int i = 0; i < NL; i++) k.l[i] = P::P8_S1[i];
    const fe inv_lazy = fe_inverse<P>(fe_add(fe_add(a, k), k));          // the same element + 16p, limbs up to 2^31
    CHECK(big_cmp(to_int<P>(inv_lazy, rinv), want) == 0, name);
  }
}

// ---- curve: affine reference over the big-int field ---------------------------------------------------------------------
struct apt { big x, y; bool inf; };
static apt a_add(const apt& P, const apt& Q, const big& q) {
  if (P.inf) return Q;
  if (Q.inf) return P;
  big lam;
  if (big_cmp(P.x, Q.x) == 0) {
    if (big_cmp(P.y, Q.y) != 0 || big_cmp(P.y, big_zero()) == 0) return apt{big_zero(), big_zero(), true};
    big three = big_zero(); three.w[0] = 3;
    lam = mulmod(mulmod(three, mulmod(P.x, P.x, q), q), invmod(addmod(P.y, P.y, q), q), q);
  } else lam = mulmod(submod(Q.y, P.y, q), invmod(submod(Q.x, P.x, q), q), q);
  big x3 = submod(submod(mulmod(lam, lam, q), P.x, q), Q.x, q);
  return apt{x3, submod(mulmod(lam, submod(P.x, x3, q), q), P.y, q), false};
}
static apt xyzz_to_affine(const xyzz& a, const big& q, const big& rinv) {
  if (xyzz_is_identity(a)) return apt{big_zero(), big_zero(), true};
  big X = to_int<Fq>(a.X, rinv), Y = to_int<Fq>(a.Y, rinv), ZZ = to_int<Fq>(a.ZZ, rinv), ZZZ = to_int<Fq>(a.ZZZ, rinv);
  return apt{mulmod(X, invmod(ZZ, q), q), mulmod(Y, invmod(ZZZ, q), q), false};
}
static bool same(const apt& a, const apt& b) { return a.inf == b.inf && (a.inf || (big_cmp(a.x, b.x) == 0 && big_cmp(a.y, b.y) == 0)); }
static bool stored_ok(const xyzz& a, const big& q) {   // the stored-point invariant of ec.hpp
  return is_N(a.X) && is_N(a.Y) && is_N(a.ZZ) && is_N(a.ZZZ) && below(a.X, 9, q) && below(a.Y, 5, q) && below(a.ZZ, 2, q) && below(a.ZZZ, 2, q);
}

static void curve_tests() {
  const big q = modulus<Fq>(), rinv = invmod(R261<Fq>(), q);
  apt G; G.x = big_zero(); G.x.w[0] = 1; G.y = big_zero(); G.y.w[0] = 2; G.inf = false;
  std::vector<apt> pts;
  apt cur = G;
  for (int i = 0; i < 40; i++) { pts.push_back(cur); cur = a_add(cur, (i % 3 == 0) ? cur : G, q); }
  auto lazy_in = [&](const big& v) { fe t = to_fe<Fq>(v); fe k; for (int i = 0; i < NL; i++) k.l[i] = Fq::P32_S1[i]; return fe_norm(fe_add(t, k)); };   // same element, value + 32p: as unreduced as a lazy external load
  for (int it = 0; it < 600; it++) {
    // a chain of mixed additions with occasional repeats (doubling path) and negations (cancellation path)
    xyzz acc = xyzz_identity();
    apt ref{big_zero(), big_zero(), true};
    int len = 1 + (int)(rnd() % 12);
    for (int j = 0; j < len; j++) {
      apt P = pts[rnd() % pts.size()];
      uint64_t mode = rnd() % 8;
      if (mode == 0 && !ref.inf) P = ref;                                   // acc += acc  -> doubling inside madd
      if (mode == 1 && !ref.inf) { P = ref; P.y = submod(big_zero(), P.y, q); }   // acc += -acc -> identity
      fe x2 = (j & 1) ? lazy_in(P.x) : to_fe<Fq>(P.x), y2 = (j & 1) ? lazy_in(P.y) : to_fe<Fq>(P.y);
      if (mode == 2) { y2 = fe_neg_red(to_fe<Fq>(P.y), Fq::P64_S1); P.y = submod(big_zero(), P.y, q); x2 = to_fe<Fq>(P.x); }   // negated load
      xyzz_madd(acc, x2, y2);
      ref = a_add(ref, P, q);
      CHECK(xyzz_is_identity(acc) || stored_ok(acc, q), "madd stored-point invariant");
      CHECK(same(xyzz_to_affine(acc, q, rinv), ref), "madd value");
    }
    // general addition / doubling of two accumulated points
    xyzz other = xyzz_identity();
    apt oref{big_zero(), big_zero(), true};
    int len2 = (int)(rnd() % 5);
    for (int j = 0; j < len2; j++) { apt P = pts[rnd() % pts.size()]; xyzz_madd(other, to_fe<Fq>(P.x), to_fe<Fq>(P.y)); oref = a_add(oref, P, q); }
    if (rnd() % 6 == 0) { other = acc; oref = ref; }
    xyzz sum = xyzz_add(acc, other), dbl = xyzz_dbl(acc);
    CHECK(same(xyzz_to_affine(sum, q, rinv), a_add(ref, oref, q)) && (xyzz_is_identity(sum) || stored_ok(sum, q)), "xyzz_add");
    CHECK(same(xyzz_to_affine(dbl, q, rinv), a_add(ref, ref, q)) && (xyzz_is_identity(dbl) || stored_ok(dbl, q)), "xyzz_dbl");
  }
}

int main() {
  field_tests<FqParams>("Fq");
  field_tests<FrParams>("Fr");
  curve_tests();
  printf(fails ? "FAILED: %d checks\n" : "fp29/ec host check OK\n", fails);
  return fails ? 1 : 0;
}
