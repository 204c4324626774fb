/* Sanitizer driver for oracle/cpu_ref.c (built with -fsanitize=address,undefined by tests/test_host_arith.py): runs the threaded
 * MSM, both FFT code paths and the row-a7 functions on small inputs and cross-checks two identities, so that memory errors or
 * undefined behaviour in the checker itself cannot hide behind "the numbers matched". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
int ref_best_multiexp(const u64*, const u64*, size_t, int, u64*);
int ref_best_fft(u64*, const u64*, uint32_t, int);
void ref_gen_scalars(u64, size_t, int, u64*);
void ref_gen_bases(u64, size_t, u64*, u64*, u64*);
void ref_expected_scalar(const u64*, size_t, const u64*, const u64*, u64*);
void ref_scalar_mul(const u64*, const u64*, u64*);
void ref_jac_to_affine(const u64*, u64*);
void ref_eval_polynomial(const u64*, size_t, const u64*, u64*);
void ref_kate_division(const u64*, size_t, const u64*, u64*);
void ref_batch_invert(u64*, size_t);
void ref_prefix_product(const u64*, size_t, u64*);
void ref_field_op(int, int, const u64*, const u64*, u64*, size_t);

int main(void) {
  int bad = 0;
  for (size_t n = 1; n <= 600; n = n * 3 + 1) {
    u64 *sc = malloc(n * 32), *bs = malloc(n * 64), t0[4], d[4], out[12], aff[8], e[4], exp_j[12], exp_a[8];
    ref_gen_bases(11 + n, n, bs, t0, d);
    ref_gen_scalars(12 + n, n, (int)(n & 1), sc);
    for (int threads = 1; threads <= 5; threads += 2) {
      ref_best_multiexp(sc, bs, n, threads, out);
      ref_jac_to_affine(out, aff);
      ref_expected_scalar(sc, n, t0, d, e);
      const u64 G[8] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL,
                        0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};   /* (1, 2) in Montgomery form */
      ref_scalar_mul(e, G, exp_j);
      ref_jac_to_affine(exp_j, exp_a);
      if (memcmp(aff, exp_a, 64)) { printf("msm mismatch n=%zu threads=%d\n", n, threads); bad++; }
    }
    free(sc); free(bs);
  }
  for (uint32_t L = 0; L <= 9; L++) {
    size_t n = (size_t)1 << L;
    u64 *a = malloc(n * 32), *b = malloc(n * 32), *q = malloc(n * 32), x[4], ev[4];
    ref_gen_scalars(77 + L, n, 0, a);
    memcpy(b, a, n * 32);
    ref_gen_scalars(99, 1, 0, x);
    ref_best_fft(a, x, L, 1);          /* any multiplier exercises the butterflies; only memory safety is checked here */
    ref_best_fft(b, x, L, 8);
    if (memcmp(a, b, n * 32)) { printf("fft serial/recursive mismatch L=%u\n", L); bad++; }
    ref_eval_polynomial(a, n, x, ev);
    ref_kate_division(a, n, x, q);
    ref_prefix_product(a, n, q);
    ref_batch_invert(a, n);
    ref_field_op(1, 0, a, b, q, n);
    free(a); free(b); free(q);
  }
  printf(bad ? "oracle sanitize: %d mismatches\n" : "oracle sanitize OK\n", bad);
  return bad != 0;
}
