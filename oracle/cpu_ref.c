/* CPU restatement of the reference's MSM / NTT path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this file's library
 * (oracle/libcpu_ref.so); the product (libzkhip.so) never links or calls it.
 *
 * PARITY UNPINNED by the reference: /root/reference holds no golden vectors at this boundary and the arithmetic
 * lives in un-vendored, un-pinned crates (halo2-axiom `halo2_proofs/src/arithmetic.rs`, `poly/domain.rs`;
 * halo2curves-axiom bn256) reached from /root/reference/aggregator/src/wrapper.rs:129 `create_proof`
 * (aggregator/Cargo.toml:7-21 names branches only).  This file restates the *published* algorithms of those
 * crates [DEP] with the same structure, so it is a fair "reference CPU path" to time:
 *   - field: 4 x u64 Montgomery (R = 2^256), CIOS multiply                      [halo2curves bn256::{Fq,Fr}]
 *   - best_multiexp: split into `threads` contiguous chunks, multiexp_serial per chunk (unsigned windows,
 *     c = 1 / 3 / ceil(ln n), segments = 256/c + 1, buckets None/Affine/Projective, running-sum), fold partials
 *   - best_fft: bit-reverse swap, serial twiddle table w^0..w^(n/2-1), serial radix-2 when log_n <= log2(threads)
 *     else recursive_butterfly_arithmetic with a two-way join per level down to the thread budget
 *   - EvaluationDomain passes: ifft divisor, distribute_powers_zeta, divide_by_vanishing_poly
 * It is pinned by tests/test_oracle.py against oracle/bn254.py (independent big-integer model) and public KATs.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct {
  u64 p[4];
  u64 inv;    /* -p^-1 mod 2^64 */
  u64 one[4]; /* R mod p */
  u64 r2[4];  /* R^2 mod p */
} field_t;

static const field_t FQ = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                           0x87d20782e4866389ULL,
                           {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
                           {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
static const field_t FR = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                           0xc2e1f593efffffffULL,
                           {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
                           {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

/* ---------------------------------------------------------------- field */
static inline int geq(const u64 a[4], const u64 b[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static inline void sub_nb(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
}
static inline void f_add(const field_t* F, u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 carry = 0, t[4];
  for (int i = 0; i < 4; i++) {
    u128 s = (u128)a[i] + b[i] + carry;
    t[i] = (u64)s;
    carry = (u64)(s >> 64);
  }
  if (carry || geq(t, F->p)) sub_nb(r, t, F->p); else memcpy(r, t, 32);
}
static inline void f_sub(const field_t* F, u64 r[4], const u64 a[4], const u64 b[4]) {
  if (geq(a, b)) { sub_nb(r, a, b); return; }
  u64 t[4];
  sub_nb(t, b, a);
  sub_nb(r, F->p, t);
}
static inline void f_mul(const field_t* F, u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u64 carry = 0;
    for (int j = 0; j < 4; j++) {
      u128 s = (u128)a[j] * b[i] + t[j] + carry;
      t[j] = (u64)s;
      carry = (u64)(s >> 64);
    }
    u128 s = (u128)t[4] + carry;
    t[4] = (u64)s;
    t[5] = (u64)(s >> 64);
    u64 m = t[0] * F->inv;
    s = (u128)m * F->p[0] + t[0];
    carry = (u64)(s >> 64);
    for (int j = 1; j < 4; j++) {
      s = (u128)m * F->p[j] + t[j] + carry;
      t[j - 1] = (u64)s;
      carry = (u64)(s >> 64);
    }
    s = (u128)t[4] + carry;
    t[3] = (u64)s;
    t[4] = t[5] + (u64)(s >> 64);
  }
  if (t[4] || geq(t, F->p)) sub_nb(r, t, F->p); else memcpy(r, t, 32);
}
static inline void f_sqr(const field_t* F, u64 r[4], const u64 a[4]) { f_mul(F, r, a, a); }
static inline int f_is_zero(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static void f_from_mont(const field_t* F, u64 r[4], const u64 a[4]) {
  const u64 one[4] = {1, 0, 0, 0};
  f_mul(F, r, a, one);
}
static void f_to_mont(const field_t* F, u64 r[4], const u64 a[4]) { f_mul(F, r, a, F->r2); }
static void f_pow(const field_t* F, u64 r[4], const u64 a[4], const u64 e[4]) {
  u64 acc[4], base[4];
  memcpy(acc, F->one, 32);
  memcpy(base, a, 32);
  for (int i = 0; i < 256; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, acc, acc, base);
    f_sqr(F, base, base);
  }
  memcpy(r, acc, 32);
}
static void f_inv(const field_t* F, u64 r[4], const u64 a[4]) {
  u64 e[4];
  const u64 two[4] = {2, 0, 0, 0};
  sub_nb(e, F->p, two);
  f_pow(F, r, a, e);
}

/* ---------------------------------------------------------------- G1 (Jacobian, a = 0) */
typedef struct { u64 x[4], y[4]; } aff_t;          /* (0,0) = identity */
typedef struct { u64 x[4], y[4], z[4]; } jac_t;    /* z = 0 identity */

static inline int aff_is_id(const aff_t* a) { return f_is_zero(a->x) && f_is_zero(a->y); }
static inline void jac_set_id(jac_t* r) { memset(r, 0, sizeof(*r)); memcpy(r->y, FQ.one, 32); }

static void jac_double(jac_t* r, const jac_t* p) { /* dbl-2009-l */
  if (f_is_zero(p->z)) { *r = *p; return; }
  u64 A[4], B[4], C[4], D[4], E[4], Fv[4], t[4];
  f_sqr(&FQ, A, p->x);
  f_sqr(&FQ, B, p->y);
  f_sqr(&FQ, C, B);
  f_add(&FQ, t, p->x, B);
  f_sqr(&FQ, t, t);
  f_sub(&FQ, t, t, A);
  f_sub(&FQ, t, t, C);
  f_add(&FQ, D, t, t);
  f_add(&FQ, E, A, A);
  f_add(&FQ, E, E, A);
  f_sqr(&FQ, Fv, E);
  u64 z3[4];
  f_mul(&FQ, z3, p->y, p->z);
  f_add(&FQ, z3, z3, z3);
  f_sub(&FQ, t, Fv, D);
  f_sub(&FQ, r->x, t, D);
  f_sub(&FQ, t, D, r->x);
  f_mul(&FQ, t, E, t);
  u64 c8[4];
  f_add(&FQ, c8, C, C);
  f_add(&FQ, c8, c8, c8);
  f_add(&FQ, c8, c8, c8);
  f_sub(&FQ, r->y, t, c8);
  memcpy(r->z, z3, 32);
}

static void jac_add(jac_t* r, const jac_t* p, const jac_t* q) { /* add-2007-bl */
  if (f_is_zero(p->z)) { *r = *q; return; }
  if (f_is_zero(q->z)) { *r = *p; return; }
  u64 z1z1[4], z2z2[4], u1[4], u2[4], s1[4], s2[4], t[4];
  f_sqr(&FQ, z1z1, p->z);
  f_sqr(&FQ, z2z2, q->z);
  f_mul(&FQ, u1, p->x, z2z2);
  f_mul(&FQ, u2, q->x, z1z1);
  f_mul(&FQ, t, q->z, z2z2);
  f_mul(&FQ, s1, p->y, t);
  f_mul(&FQ, t, p->z, z1z1);
  f_mul(&FQ, s2, q->y, t);
  if (memcmp(u1, u2, 32) == 0) {
    if (memcmp(s1, s2, 32) == 0) { jac_double(r, p); return; }
    jac_set_id(r);
    return;
  }
  u64 h[4], i[4], j[4], rr[4], v[4];
  f_sub(&FQ, h, u2, u1);
  f_add(&FQ, i, h, h);
  f_sqr(&FQ, i, i);
  f_mul(&FQ, j, h, i);
  f_sub(&FQ, rr, s2, s1);
  f_add(&FQ, rr, rr, rr);
  f_mul(&FQ, v, u1, i);
  jac_t o;
  f_sqr(&FQ, o.x, rr);
  f_sub(&FQ, o.x, o.x, j);
  f_sub(&FQ, o.x, o.x, v);
  f_sub(&FQ, o.x, o.x, v);
  f_sub(&FQ, t, v, o.x);
  f_mul(&FQ, o.y, rr, t);
  f_mul(&FQ, t, s1, j);
  f_add(&FQ, t, t, t);
  f_sub(&FQ, o.y, o.y, t);
  f_add(&FQ, t, p->z, q->z);
  f_sqr(&FQ, t, t);
  f_sub(&FQ, t, t, z1z1);
  f_sub(&FQ, t, t, z2z2);
  f_mul(&FQ, o.z, t, h);
  *r = o;
}

static void jac_add_mixed(jac_t* r, const jac_t* p, const aff_t* q) { /* madd-2007-bl */
  if (aff_is_id(q)) { *r = *p; return; }
  if (f_is_zero(p->z)) {
    memcpy(r->x, q->x, 32); memcpy(r->y, q->y, 32); memcpy(r->z, FQ.one, 32);
    return;
  }
  u64 z1z1[4], u2[4], s2[4], t[4];
  f_sqr(&FQ, z1z1, p->z);
  f_mul(&FQ, u2, q->x, z1z1);
  f_mul(&FQ, t, p->z, z1z1);
  f_mul(&FQ, s2, q->y, t);
  if (memcmp(p->x, u2, 32) == 0) {
    if (memcmp(p->y, s2, 32) == 0) { jac_double(r, p); return; }
    jac_set_id(r);
    return;
  }
  u64 h[4], hh[4], i[4], j[4], rr[4], v[4];
  f_sub(&FQ, h, u2, p->x);
  f_sqr(&FQ, hh, h);
  f_add(&FQ, i, hh, hh);
  f_add(&FQ, i, i, i);
  f_mul(&FQ, j, h, i);
  f_sub(&FQ, rr, s2, p->y);
  f_add(&FQ, rr, rr, rr);
  f_mul(&FQ, v, p->x, i);
  jac_t o;
  f_sqr(&FQ, o.x, rr);
  f_sub(&FQ, o.x, o.x, j);
  f_sub(&FQ, o.x, o.x, v);
  f_sub(&FQ, o.x, o.x, v);
  f_sub(&FQ, t, v, o.x);
  f_mul(&FQ, o.y, rr, t);
  f_mul(&FQ, t, p->y, j);
  f_add(&FQ, t, t, t);
  f_sub(&FQ, o.y, o.y, t);
  f_add(&FQ, t, p->z, h);
  f_sqr(&FQ, t, t);
  f_sub(&FQ, t, t, z1z1);
  f_sub(&FQ, o.z, t, hh);
  *r = o;
}

static void jac_to_affine(aff_t* r, const jac_t* p) {
  if (f_is_zero(p->z)) { memset(r, 0, sizeof(*r)); return; }
  u64 zi[4], zi2[4], zi3[4];
  f_inv(&FQ, zi, p->z);
  f_sqr(&FQ, zi2, zi);
  f_mul(&FQ, zi3, zi2, zi);
  f_mul(&FQ, r->x, p->x, zi2);
  f_mul(&FQ, r->y, p->y, zi3);
}

/* ---------------------------------------------------------------- multiexp_serial / best_multiexp */
static int window_bits(size_t n) {
  if (n < 4) return 1;
  if (n < 32) return 3;
  return (int)ceil(log((double)n));
}

static inline size_t get_at(size_t segment, int c, const unsigned char repr[32]) {
  size_t skip_bits = segment * (size_t)c, skip_bytes = skip_bits / 8;
  if (skip_bytes >= 32) return 0;
  unsigned char v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = repr[skip_bytes + i];
  u64 tmp;
  memcpy(&tmp, v, 8);
  tmp >>= skip_bits - skip_bytes * 8;
  tmp %= (u64)1 << c;
  return (size_t)tmp;
}

typedef struct { int tag; /* 0 None, 1 Affine, 2 Projective */ aff_t a; jac_t j; } bucket_t;

static void multiexp_serial(const u64* scalars, const aff_t* bases, size_t n, jac_t* acc) {
  unsigned char* reprs = (unsigned char*)malloc(n * 32 + 32);
  for (size_t i = 0; i < n; i++) {
    u64 canon[4];
    f_from_mont(&FR, canon, scalars + 4 * i); /* `to_repr()` */
    memcpy(reprs + 32 * i, canon, 32);
  }
  int c = window_bits(n);
  size_t segments = 256 / (size_t)c + 1, nb = ((size_t)1 << c) - 1;
  bucket_t* buckets = (bucket_t*)malloc(nb * sizeof(bucket_t));
  for (size_t seg = segments; seg-- > 0;) {
    for (int i = 0; i < c; i++) jac_double(acc, acc);
    for (size_t b = 0; b < nb; b++) buckets[b].tag = 0;
    for (size_t i = 0; i < n; i++) {
      size_t d = get_at(seg, c, reprs + 32 * i);
      if (d == 0) continue;
      bucket_t* bk = &buckets[d - 1];
      if (bk->tag == 0) { bk->tag = 1; bk->a = bases[i]; }
      else if (bk->tag == 1) {
        jac_t t;
        if (aff_is_id(&bk->a)) jac_set_id(&t); else { memcpy(t.x, bk->a.x, 32); memcpy(t.y, bk->a.y, 32); memcpy(t.z, FQ.one, 32); }
        jac_add_mixed(&bk->j, &t, &bases[i]);
        bk->tag = 2;
      } else jac_add_mixed(&bk->j, &bk->j, &bases[i]);
    }
    jac_t running;
    jac_set_id(&running);
    for (size_t b = nb; b-- > 0;) {
      bucket_t* bk = &buckets[b];
      if (bk->tag == 1) jac_add_mixed(&running, &running, &bk->a);
      else if (bk->tag == 2) jac_add(&running, &running, &bk->j);
      jac_add(acc, acc, &running);
    }
  }
  free(buckets);
  free(reprs);
}

typedef struct { const u64* scalars; const aff_t* bases; size_t n; jac_t acc; } msm_job_t;
static void* msm_worker(void* arg) {
  msm_job_t* j = (msm_job_t*)arg;
  jac_set_id(&j->acc);
  multiexp_serial(j->scalars, j->bases, j->n, &j->acc);
  return NULL;
}

int ref_best_multiexp(const u64* scalars, const u64* bases, size_t n, int threads, u64 out_xyz[12]) {
  jac_t acc;
  jac_set_id(&acc);
  if (threads < 1) threads = 1;
  if (n > (size_t)threads) {
    size_t chunk = n / (size_t)threads, nchunks = (n + chunk - 1) / chunk;
    msm_job_t* jobs = (msm_job_t*)malloc(nchunks * sizeof(msm_job_t));
    pthread_t* th = (pthread_t*)malloc(nchunks * sizeof(pthread_t));
    for (size_t k = 0; k < nchunks; k++) {
      size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
      jobs[k].scalars = scalars + 4 * lo;
      jobs[k].bases = (const aff_t*)bases + lo;
      jobs[k].n = hi - lo;
      pthread_create(&th[k], NULL, msm_worker, &jobs[k]);
    }
    for (size_t k = 0; k < nchunks; k++) {
      pthread_join(th[k], NULL);
      jac_add(&acc, &acc, &jobs[k].acc);
    }
    free(jobs);
    free(th);
  } else {
    multiexp_serial(scalars, (const aff_t*)bases, n, &acc);
  }
  memcpy(out_xyz, &acc, 96);
  return 0;
}

/* ---------------------------------------------------------------- best_fft */
static size_t bitreverse(size_t n, unsigned l) {
  size_t r = 0;
  for (unsigned i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
  return r;
}

typedef struct { u64* a; size_t n, twiddle_chunk; const u64* tw; int depth; } fft_job_t;

static void butterfly_rec(u64* a, size_t n, size_t twiddle_chunk, const u64* tw, int depth);
static void* fft_worker(void* arg) {
  fft_job_t* j = (fft_job_t*)arg;
  butterfly_rec(j->a, j->n, j->twiddle_chunk, j->tw, j->depth);
  return NULL;
}

static void butterfly_rec(u64* a, size_t n, size_t twiddle_chunk, const u64* tw, int depth) {
  if (n == 2) {
    u64 t[4];
    memcpy(t, a + 4, 32);
    memcpy(a + 4, a, 32);
    f_add(&FR, a, a, t);
    f_sub(&FR, a + 4, a + 4, t);
    return;
  }
  u64 *left = a, *right = a + 4 * (n / 2);
  if (depth > 0) { /* `multicore::join` */
    fft_job_t job = {right, n / 2, twiddle_chunk * 2, tw, depth - 1};
    pthread_t th;
    pthread_create(&th, NULL, fft_worker, &job);
    butterfly_rec(left, n / 2, twiddle_chunk * 2, tw, depth - 1);
    pthread_join(th, NULL);
  } else {
    butterfly_rec(left, n / 2, twiddle_chunk * 2, tw, 0);
    butterfly_rec(right, n / 2, twiddle_chunk * 2, tw, 0);
  }
  u64 t[4];
  memcpy(t, right, 32);
  memcpy(right, left, 32);
  f_add(&FR, left, left, t);
  f_sub(&FR, right, right, t);
  for (size_t i = 1; i < n / 2; i++) {
    u64* x = left + 4 * i;
    u64* y = right + 4 * i;
    f_mul(&FR, t, y, tw + 4 * (i * twiddle_chunk));
    memcpy(y, x, 32);
    f_add(&FR, x, x, t);
    f_sub(&FR, y, y, t);
  }
}

int ref_best_fft(u64* a, const u64 omega[4], uint32_t log_n, int threads) {
  size_t n = (size_t)1 << log_n;
  if (threads < 1) threads = 1;
  int log_threads = 0;
  while ((1 << (log_threads + 1)) <= threads) log_threads++;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitreverse(k, log_n);
    if (k < rk) {
      u64 t[4];
      memcpy(t, a + 4 * k, 32);
      memcpy(a + 4 * k, a + 4 * rk, 32);
      memcpy(a + 4 * rk, t, 32);
    }
  }
  if (n < 2) return 0;
  u64* tw = (u64*)malloc((n / 2) * 32);
  u64 w[4];
  memcpy(w, FR.one, 32);
  for (size_t i = 0; i < n / 2; i++) {
    memcpy(tw + 4 * i, w, 32);
    f_mul(&FR, w, w, omega);
  }
  if ((int)log_n <= log_threads) {
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t s = 0; s < log_n; s++) {
      for (size_t base = 0; base < n; base += chunk) {
        u64 *left = a + 4 * base, *right = left + 4 * (chunk / 2), t[4];
        memcpy(t, right, 32);
        memcpy(right, left, 32);
        f_add(&FR, left, left, t);
        f_sub(&FR, right, right, t);
        for (size_t i = 1; i < chunk / 2; i++) {
          u64 *x = left + 4 * i, *y = right + 4 * i;
          f_mul(&FR, t, y, tw + 4 * (i * twiddle_chunk));
          memcpy(y, x, 32);
          f_add(&FR, x, x, t);
          f_sub(&FR, y, y, t);
        }
      }
      chunk *= 2;
      twiddle_chunk /= 2;
    }
  } else {
    butterfly_rec(a, n, 1, tw, log_threads);
  }
  free(tw);
  return 0;
}

/* ---------------------------------------------------------------- EvaluationDomain passes */
void ref_scale(u64* a, size_t n, const u64 c[4]) { /* ifft divisor loop */
  for (size_t i = 0; i < n; i++) f_mul(&FR, a + 4 * i, a + 4 * i, c);
}
void ref_distribute_powers_zeta(u64* a, size_t n, const u64 p1[4], const u64 p2[4]) {
  for (size_t i = 0; i < n; i++) {
    size_t k = i % 3;
    if (k == 1) f_mul(&FR, a + 4 * i, a + 4 * i, p1);
    else if (k == 2) f_mul(&FR, a + 4 * i, a + 4 * i, p2);
  }
}
void ref_mul_periodic(u64* a, size_t n, const u64* table, size_t period) { /* divide_by_vanishing_poly */
  for (size_t i = 0; i < n; i++) f_mul(&FR, a + 4 * i, a + 4 * i, table + 4 * (i % period));
}

/* ---------------------------------------------------------------- row a7 (serial restatements) */
void ref_eval_polynomial(const u64* poly, size_t n, const u64 point[4], u64 out[4]) { /* `evaluate`: Horner from the top */
  u64 acc[4] = {0, 0, 0, 0};
  for (size_t i = n; i-- > 0;) {
    f_mul(&FR, acc, acc, point);
    f_add(&FR, acc, acc, poly + 4 * i);
  }
  memcpy(out, acc, 32);
}
/* `eval_polynomial` as halo2-axiom runs it on a multi-core host [DEP arithmetic.rs]: the coefficients are cut into `threads`
 * contiguous chunks, each chunk is evaluated by Horner (`evaluate`) and scaled by point^(chunk start), the parts are summed.
 * Same value as the serial form; used by the large-size NTT spot checks (a(w^i) for sampled i) and timed in bench.py. */
typedef struct { const u64* poly; size_t n; const u64* point; size_t start; u64 out[4]; } evalp_job_t;
static void* evalp_worker(void* arg) {
  evalp_job_t* j = (evalp_job_t*)arg;
  u64 acc[4], e[4] = {(u64)j->start, 0, 0, 0}, pw[4];
  ref_eval_polynomial(j->poly, j->n, j->point, acc);
  f_pow(&FR, pw, j->point, e);
  f_mul(&FR, j->out, acc, pw);
  return NULL;
}
void ref_eval_polynomial_mt(const u64* poly, size_t n, const u64 point[4], int threads, u64 out[4]) {
  if (threads < 1) threads = 1;
  if (n < 2 * (size_t)threads || threads == 1) { ref_eval_polynomial(poly, n, point, out); return; }
  size_t chunk = n / (size_t)threads, nchunks = (n + chunk - 1) / chunk;
  evalp_job_t* jobs = (evalp_job_t*)malloc(nchunks * sizeof(evalp_job_t));
  pthread_t* th = (pthread_t*)malloc(nchunks * sizeof(pthread_t));
  for (size_t k = 0; k < nchunks; k++) {
    size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
    jobs[k].poly = poly + 4 * lo; jobs[k].n = hi - lo; jobs[k].point = point; jobs[k].start = lo;
    pthread_create(&th[k], NULL, evalp_worker, &jobs[k]);
  }
  u64 acc[4] = {0, 0, 0, 0};
  for (size_t k = 0; k < nchunks; k++) {
    pthread_join(th[k], NULL);
    f_add(&FR, acc, acc, jobs[k].out);
  }
  memcpy(out, acc, 32);
  free(jobs);
  free(th);
}
void ref_kate_division(const u64* a, size_t n, const u64 b[4], u64* q) { /* b = -b; q[i-1] = a[i] - tmp; tmp = q[i-1] * b */
  if (n < 2) return;
  u64 nb[4], zero[4] = {0, 0, 0, 0}, tmp[4] = {0, 0, 0, 0};
  f_sub(&FR, nb, zero, b);
  for (size_t i = n - 1; i >= 1; i--) {
    u64 lead[4];
    f_sub(&FR, lead, a + 4 * i, tmp);
    memcpy(q + 4 * (i - 1), lead, 32);
    f_mul(&FR, tmp, lead, nb);
  }
}
void ref_batch_invert(u64* a, size_t n) { /* ff::BatchInvert: zeros are skipped and stay zero */
  u64* pref = (u64*)malloc((n + 1) * 32);
  u64 acc[4];
  memcpy(acc, FR.one, 32);
  for (size_t i = 0; i < n; i++) {
    memcpy(pref + 4 * i, acc, 32);
    if (!f_is_zero(a + 4 * i)) f_mul(&FR, acc, acc, a + 4 * i);
  }
  u64 inv[4];
  f_inv(&FR, inv, acc);
  for (size_t i = n; i-- > 0;) {
    if (f_is_zero(a + 4 * i)) continue;
    u64 t[4];
    f_mul(&FR, t, inv, pref + 4 * i);
    f_mul(&FR, inv, inv, a + 4 * i);
    memcpy(a + 4 * i, t, 32);
  }
  free(pref);
}
void ref_prefix_product(const u64* v, size_t n, u64* out) { /* z[0] = 1; z[i+1] = z[i] * v[i] */
  u64 cur[4];
  memcpy(cur, FR.one, 32);
  for (size_t i = 0; i < n; i++) {
    u64 t[4];
    memcpy(t, v + 4 * i, 32);
    memcpy(out + 4 * i, cur, 32);
    f_mul(&FR, cur, cur, t);
  }
}

/* ---------------------------------------------------------------- elementwise hooks + generators */
void ref_field_op(int field, int op, const u64* a, const u64* b, u64* out, size_t n) {
  const field_t* F = field == 0 ? &FQ : &FR;
  for (size_t i = 0; i < n; i++) {
    if (op == 0) f_mul(F, out + 4 * i, a + 4 * i, b + 4 * i);
    else if (op == 1) f_add(F, out + 4 * i, a + 4 * i, b + 4 * i);
    else if (op == 2) f_sub(F, out + 4 * i, a + 4 * i, b + 4 * i);
    else f_sqr(F, out + 4 * i, a + 4 * i);
  }
}

void ref_jac_to_affine(const u64 xyz[12], u64 out[8]) { jac_to_affine((aff_t*)out, (const jac_t*)xyz); }
void ref_jac_add(const u64 a[12], const u64 b[12], u64 out[12]) { jac_t r; jac_add(&r, (const jac_t*)a, (const jac_t*)b); memcpy(out, &r, 96); }

/* k * P, k given as canonical (non-Montgomery) little-endian limbs */
void ref_scalar_mul(const u64 k[4], const u64 base[8], u64 out_xyz[12]) {
  jac_t acc, b;
  jac_set_id(&acc);
  const aff_t* P = (const aff_t*)base;
  if (aff_is_id(P)) { memcpy(out_xyz, &acc, 96); return; }
  memcpy(b.x, P->x, 32); memcpy(b.y, P->y, 32); memcpy(b.z, FQ.one, 32);
  for (int i = 255; i >= 0; i--) {
    jac_double(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) jac_add(&acc, &acc, &b);
  }
  memcpy(out_xyz, &acc, 96);
}

static u64 splitmix(u64* s) {
  *s += 0x9E3779B97F4A7C15ULL;
  u64 z = *s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void gen_fr_canon(u64* s, u64 out[4]) { /* 256 random bits mod r, identical to oracle/bn254.py SplitMix64.fr */
  u64 v[4];
  for (int i = 0; i < 4; i++) v[i] = splitmix(s);
  while (geq(v, FR.p)) sub_nb(v, v, FR.p);
  memcpy(out, v, 32);
}

/* kind 0: uniform Fr; kind 1: witness-like (60 % zero, 30 % < 2^88, 10 % uniform).  Output: Montgomery limbs. */
void ref_gen_scalars(u64 seed, size_t n, int kind, u64* out) {
  u64 s = seed;
  for (size_t i = 0; i < n; i++) {
    u64 v[4];
    if (kind == 1) {
      u64 sel = splitmix(&s) % 10;
      gen_fr_canon(&s, v);
      if (sel < 6) memset(v, 0, 32);
      else if (sel < 9) { v[1] &= ((u64)1 << 24) - 1; v[2] = 0; v[3] = 0; }
    } else gen_fr_canon(&s, v);
    f_to_mont(&FR, out + 4 * i, v);
  }
}

/* bases[i] = (t0 + i d) * G, affine Montgomery; returns canonical t0, d (so that MSM(a, bases) = [sum a_i (t0 + i d)] G) */
void ref_gen_bases(u64 seed, size_t n, u64* out, u64 t0[4], u64 d[4]) {
  u64 s = seed;
  gen_fr_canon(&s, t0);
  gen_fr_canon(&s, d);
  aff_t G;
  const u64 one[4] = {1, 0, 0, 0}, two[4] = {2, 0, 0, 0};
  f_to_mont(&FQ, G.x, one);
  f_to_mont(&FQ, G.y, two);
  jac_t P, D;
  ref_scalar_mul(t0, (const u64*)&G, (u64*)&P);
  ref_scalar_mul(d, (const u64*)&G, (u64*)&D);
  jac_t* js = (jac_t*)malloc((n + 1) * sizeof(jac_t));
  for (size_t i = 0; i < n; i++) { js[i] = P; jac_add(&P, &P, &D); }
  /* batch normalisation (Montgomery trick) */
  u64* prefix = (u64*)malloc((n + 1) * 32);
  u64 acc[4];
  memcpy(acc, FQ.one, 32);
  for (size_t i = 0; i < n; i++) {
    memcpy(prefix + 4 * i, acc, 32);
    if (!f_is_zero(js[i].z)) f_mul(&FQ, acc, acc, js[i].z);
  }
  u64 inv[4];
  f_inv(&FQ, inv, acc);
  aff_t* o = (aff_t*)out;
  for (size_t i = n; i-- > 0;) {
    if (f_is_zero(js[i].z)) { memset(&o[i], 0, sizeof(aff_t)); continue; }
    u64 zi[4], zi2[4], zi3[4];
    f_mul(&FQ, zi, inv, prefix + 4 * i);
    f_mul(&FQ, inv, inv, js[i].z);
    f_sqr(&FQ, zi2, zi);
    f_mul(&FQ, zi3, zi2, zi);
    f_mul(&FQ, o[i].x, js[i].x, zi2);
    f_mul(&FQ, o[i].y, js[i].y, zi3);
  }
  free(prefix);
  free(js);
}

/* sum_i a_i * (t0 + i d) mod r, a in Montgomery limbs; result canonical */
void ref_expected_scalar(const u64* scalars, size_t n, const u64 t0[4], const u64 d[4], u64 out[4]) {
  u64 acc[4] = {0, 0, 0, 0}, t[4], dm[4], tm[4];
  f_to_mont(&FR, tm, t0);
  f_to_mont(&FR, dm, d);
  for (size_t i = 0; i < n; i++) {
    f_mul(&FR, t, scalars + 4 * i, tm);
    f_add(&FR, acc, acc, t);
    f_add(&FR, tm, tm, dm);
  }
  f_from_mont(&FR, out, acc);
}
