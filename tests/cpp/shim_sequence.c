/* The call sequence of the Rust shim (rust-shim/zkhip_ffi.rs, arithmetic_patch.rs, commitment_patch.rs, domain_patch.rs) over the life
 * of two `ParamsKZG` objects, issued from plain C99 through include/zkhip.h -- the same calls, in the same order, with the same pointer
 * arithmetic (`&self.g[..size]` = the registered pointer with a shorter length), that the patched halo2-axiom crate makes under the
 * reference's prover (/root/reference/aggregator/src/wrapper.rs:106-137: gen_pk, create_proof; gen_srs at
 * /root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69).  No Rust toolchain exists in the build image, so this is how the
 * shim's behaviour is exercised on the GPU (tests/test_gpu_shim_sequence.py compiles and runs it, and checks the printed commitments
 * against the oracle).
 *
 *   zkhip_ffi::usable()         init + [1] G == G layout self-test
 *   ParamsKZG::setup / read     register(g), register(g_lagrange)                       (commitment_patch.rs: zkhip_pinned)
 *   commit / commit_lagrange    msm(scalars, &g[..n']) for n' <= len                     (arithmetic_patch.rs: best_multiexp)
 *   EvaluationDomain            ntt, ifft_scaled, coeff_to_extended, extended_to_coeff, mul_periodic   (best_fft, domain_patch.rs)
 *   g_to_lagrange               zkhip_g_to_lagrange (setup / from_parts / downsize)
 *   Drop                        unregister(g), unregister(g_lagrange)
 *   a second ParamsKZG whose Vecs land on the SAME addresses, other points: register again, commit
 *
 * Every registered-path result must equal the result of the same call made while the array was not registered (general path: per-call
 * upload, per-window buckets, GLV) -- two different algorithms over different tables.  Exit status 0 = all checks passed.
 * Output: lines `commit <tag> <n> <x limbs> <y limbs>` (affine, Montgomery words) for the Python side to check against the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkhip.h"

#define LOG_N 14
#define N ((size_t)1 << LOG_N)
#define K_NTT 12
#define K_EXT 14

static const uint64_t FR_ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
static const uint64_t FQ_ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
static const uint64_t FQ_TWO[4] = {0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
/* omega_12, its inverse, 2^-12; omega_14, its inverse, 2^-14; Fr::ZETA -- Montgomery words (tests/test_gpu_shim_sequence.py re-derives them) */
static const uint64_t OMEGA[4] = {0xa3a44167563e01d9ULL, 0x2c985eba520cba25ULL, 0x630be0929f706d21ULL, 0x0f8cad63348de52eULL};
static const uint64_t OMEGA_INV[4] = {0xfa70a02c372988b4ULL, 0xf2bb7c7886a3be8fULL, 0x465ac77d9e749fe3ULL, 0x1ee54ba9f8cd0adeULL};
static const uint64_t N_INV[4] = {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0010000000000000ULL};
static const uint64_t EXT_OMEGA[4] = {0x2a136f90fe079611ULL, 0x9091953b5a8c7132ULL, 0xcddff23737965037ULL, 0x0c29a15e02149426ULL};
static const uint64_t EXT_OMEGA_INV[4] = {0x6ba35e8a5b34fc52ULL, 0xc7eed92f1738ad54ULL, 0x1b37fd9e4df55656ULL, 0x08c61bd5a1366af8ULL};
static const uint64_t EXT_DIV[4] = {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0004000000000000ULL};
static const uint64_t ZETA[4] = {0x0363f29955fcd653ULL, 0x73e7950b5fc1e200ULL, 0xc5fce83e576d9d24ULL, 0x059c805da1c3a4d4ULL};

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: %s (%s)\n", __LINE__, what, zkhip_last_error()); failures++; } } while (0)
#define OK(call) CHECK((call) == ZKHIP_OK, #call)

static uint64_t rng_state;
static uint64_t next_u64(void) {          /* xorshift64* -- the Python side regenerates the same scalars */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return rng_state * 0x2545F4914F6CDD1DULL;
}
static void fill_scalars(uint64_t *s, size_t n, uint64_t seed) {
  size_t i;
  rng_state = seed;
  for (i = 0; i < n; i++) {
    s[4 * i] = next_u64(); s[4 * i + 1] = next_u64(); s[4 * i + 2] = next_u64();
    s[4 * i + 3] = next_u64() & 0x0fffffffffffffffULL;      /* < 2^252 < r: a canonical Montgomery word pattern */
  }
}

/* bases[i] = (t0 + i d) G, built on the device (the library's synthetic SRS) and brought to host memory, as a file read would */
static void fill_bases(uint64_t *host, size_t n, uint64_t t0_small, uint64_t d_small) {
  /* walk parameters as Montgomery WORDS (t0_small, 0, 0, 0) / (d_small, 0, 0, 0): any Fr value is a valid parameter, and the Python side
   * decodes the words it is given (the value is word * 2^-256 mod r) */
  uint64_t t0[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
  void *dev = NULL;
  t0[0] = t0_small; d[0] = d_small;
  OK(zkhip_alloc(n * 64, &dev));
  OK(zkhip_g1_gen_walk_device(t0, d, n, dev, NULL));
  OK(zkhip_sync());
  OK(zkhip_download(host, dev, n * 64));
  OK(zkhip_free(dev));
}

static int same_point(const uint64_t a_xyz[12], const uint64_t b_xyz[12]) {
  uint64_t in[24], out[16];
  memcpy(in, a_xyz, 96); memcpy(in + 12, b_xyz, 96);
  if (zkhip_g1_batch_normalize(in, 2, out) != ZKHIP_OK) return 0;
  return memcmp(out, out + 8, 64) == 0;
}

static void print_commit(const char *tag, size_t n, const uint64_t xyz[12]) {
  uint64_t aff[8];
  int i;
  if (zkhip_g1_batch_normalize(xyz, 1, aff) != ZKHIP_OK) { failures++; return; }
  printf("commit %s %zu", tag, n);
  for (i = 0; i < 8; i++) printf(" %016llx", (unsigned long long)aff[i]);
  printf("\n");
}

int main(void) {
  static const size_t sizes[] = {N, N / 2, 1000, 3, 1, 0};
  uint64_t *g = malloc(N * 64), *g_lagrange = malloc(N * 64), *scalars = malloc(N * 32);
  uint64_t *poly = malloc(((size_t)1 << K_EXT) * 32), *keep = malloc(((size_t)1 << K_EXT) * 32), *ext = malloc(((size_t)1 << K_EXT) * 32);
  uint64_t ref_g[6][12], ref_l[6][12], ref_off[12], out[12], gen[8];
  size_t i;
  if (!g || !g_lagrange || !scalars || !poly || !keep || !ext) return 2;

  /* ---- zkhip_ffi::usable(): init, then [1] G must come back as G (layout of Fr / G1Affine / G1) ---- */
  OK(zkhip_init(NULL, 0));
  memcpy(gen, FQ_ONE, 32); memcpy(gen + 4, FQ_TWO, 32);
  OK(zkhip_msm_g1(FR_ONE, gen, 1, out));
  { uint64_t aff[8]; OK(zkhip_g1_batch_normalize(out, 1, aff)); CHECK(memcmp(aff, gen, 64) == 0, "[1] G == G"); }

  /* ---- first ParamsKZG: results with nothing registered (general path), then the same calls on the pinned arrays ---- */
  fill_bases(g, N, 5, 7);
  fill_bases(g_lagrange, N, 1000003, 11);
  fill_scalars(scalars, N, 0x5A4B534E41500001ULL);
  for (i = 0; i < 6; i++) {
    OK(zkhip_msm_g1(scalars, g, sizes[i], ref_g[i]));
    OK(zkhip_msm_g1(scalars, g_lagrange, sizes[i], ref_l[i]));
  }
  OK(zkhip_msm_g1(scalars + 4 * 100, g + 8 * 4096, 5000, ref_off));
  print_commit("g_unregistered", N, ref_g[0]);
  print_commit("g_lagrange_unregistered", N / 2, ref_l[1]);

  OK(zkhip_register_bases(g, N));                   /* ParamsKZG::setup / read_custom / from_parts -> zkhip_pinned() */
  OK(zkhip_register_bases(g_lagrange, N));
  for (i = 0; i < 6; i++) {                         /* commit(&poly) = best_multiexp(&scalars, &self.g[..size]) */
    OK(zkhip_msm_g1(scalars, g, sizes[i], out));
    CHECK(same_point(out, ref_g[i]), "commit on the pinned g == unregistered result");
    if (i == 0) print_commit("g_registered", N, out);
    OK(zkhip_msm_g1(scalars, g_lagrange, sizes[i], out));
    CHECK(same_point(out, ref_l[i]), "commit_lagrange on the pinned g_lagrange == unregistered result");
    if (i == 1) print_commit("g_lagrange_registered", N / 2, out);
  }
  OK(zkhip_msm_g1(scalars + 4 * 100, g + 8 * 4096, 5000, out));     /* any sub-range of a pinned array */
  CHECK(same_point(out, ref_off), "sub-range inside the pinned array");
  { uint64_t zero[12]; memset(zero, 0xff, sizeof zero); OK(zkhip_msm_g1(scalars, g, 0, zero)); CHECK(zero[8] == 0 && zero[9] == 0 && zero[10] == 0 && zero[11] == 0, "n = 0 gives the identity (z = 0)"); }

  /* ---- EvaluationDomain through best_fft and the fused prologues of domain_patch.rs ---- */
  fill_scalars(poly, (size_t)1 << K_NTT, 77);
  memcpy(keep, poly, ((size_t)1 << K_NTT) * 32);
  OK(zkhip_ntt_fr(poly, OMEGA, K_NTT));                                          /* best_fft(a, omega, k) */
  CHECK(memcmp(poly, keep, ((size_t)1 << K_NTT) * 32) != 0, "the transform changed the data");
  OK(zkhip_ifft_scaled(poly, OMEGA_INV, K_NTT, N_INV));                          /* EvaluationDomain::ifft */
  CHECK(memcmp(poly, keep, ((size_t)1 << K_NTT) * 32) == 0, "ifft(fft(a)) == a");
  OK(zkhip_coeff_to_extended(poly, K_NTT, ext, K_EXT, EXT_OMEGA, ZETA));         /* coeff_to_extended */
  {
    uint64_t table[4], *back = malloc(((size_t)3 << K_NTT) * 32);
    size_t j, tail_zero = 1;
    if (!back) return 2;
    memcpy(table, FR_ONE, 32);
    OK(zkhip_mul_periodic(ext, (size_t)1 << K_EXT, table, 1));                   /* divide_by_vanishing_poly with a table of ones */
    OK(zkhip_extended_to_coeff(ext, K_EXT, EXT_OMEGA_INV, EXT_DIV, ZETA, back, (size_t)3 << K_NTT));   /* extended_to_coeff: n (j - 1) coefficients */
    CHECK(memcmp(back, keep, ((size_t)1 << K_NTT) * 32) == 0, "extended_to_coeff(coeff_to_extended(p)) == p");
    for (j = (size_t)4 << K_NTT; j < (size_t)12 << K_NTT; j++) tail_zero &= (back[j] == 0);
    CHECK(tail_zero, "coefficients n .. 3n of a degree < n polynomial are zero");
    free(back);
  }

  /* ---- g_to_lagrange (ParamsKZG::setup / from_parts / downsize; commitment_patch.rs item 5): 2^11 Jacobian points in, affine Lagrange basis out.
   * g_lagrange[i] = (1/n) sum_j omega^(-i j) g[j], hence sum_i g_lagrange[i] = g[0]: checked with an MSM of ones over the result ---- */
  {
    const uint32_t kk = 11;
    const size_t m = (size_t)1 << kk;
    uint64_t *jac = malloc(m * 96), *lag = malloc(m * 64), *ones = malloc(m * 32), sum[12], first[12];
    size_t j;
    if (!jac || !lag || !ones) return 2;
    for (j = 0; j < m; j++) { memcpy(jac + 12 * j, g + 8 * j, 64); memcpy(jac + 12 * j + 8, FQ_ONE, 32); memcpy(ones + 4 * j, FR_ONE, 32); }   /* z = 1 */
    OK(zkhip_g_to_lagrange(jac, kk, lag));
    OK(zkhip_msm_g1(ones, lag, m, sum));
    memcpy(first, jac, 96);
    CHECK(same_point(sum, first), "sum of the Lagrange-basis points == g[0]");
    CHECK(memcmp(lag, g, 64) != 0, "the Lagrange basis differs from the monomial one");
    free(jac); free(lag); free(ones);
  }

  /* ---- Drop for ParamsKZG: unregister BEFORE the Vecs are freed ---- */
  OK(zkhip_unregister_bases(g));
  OK(zkhip_unregister_bases(g_lagrange));
  CHECK(zkhip_unregister_bases(g) != ZKHIP_OK, "a second unregister of the same address is reported, not fatal");

  /* ---- a second ParamsKZG whose allocations land on the same addresses, with other points ---- */
  fill_bases(g, N, 900001, 13);
  fill_bases(g_lagrange, N, 31, 17);
  OK(zkhip_msm_g1(scalars, g, N, ref_g[0]));                                    /* not registered any more: general path on the NEW points */
  OK(zkhip_msm_g1(scalars, g_lagrange, N / 2, ref_l[1]));
  CHECK(!same_point(ref_g[0], ref_g[1]) , "sanity: different sizes give different commitments");
  OK(zkhip_register_bases(g, N));
  OK(zkhip_register_bases(g_lagrange, N));
  OK(zkhip_msm_g1(scalars, g, N, out));
  CHECK(same_point(out, ref_g[0]), "re-registered address serves the NEW points");
  print_commit("g2_registered", N, out);
  OK(zkhip_msm_g1(scalars, g_lagrange, N / 2, out));
  CHECK(same_point(out, ref_l[1]), "re-registered g_lagrange serves the NEW points");
  OK(zkhip_unregister_bases(g));
  OK(zkhip_unregister_bases(g_lagrange));
  zkhip_shutdown();
  free(g); free(g_lagrange); free(scalars); free(poly); free(keep); free(ext);
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("shim sequence OK\n");
  return 0;
}
