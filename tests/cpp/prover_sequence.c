/* The call sequence of rust-shim/prover_patch.rs for ONE proof of a halo2-lib shaped circuit, issued from plain C99 through
 * include/zkhip.h, three ways over the same witness:
 *
 *   H  host-buffer, call by call     what an unmodified create_proof makes through arithmetic_patch.rs / domain_patch.rs: one zkhip_msm_g1 per
 *                                    column, one zkhip_ifft_scaled / zkhip_coeff_to_extended per column, host-buffer row programs
 *   B  host-buffer, one call per phase   prover_patch.rs mode (a): zkhip_msm_g1_batch, zkhip_ifft_scaled_batch, zkhip_coeff_to_extended_batch
 *   D  device-resident               prover_patch.rs mode (b): the columns are zkhip_alloc'd handles from the upload of the witness to the
 *                                    quotient commitments and the multi-open argument (SHPLONK on the device-resident polynomials); only
 *                                    commitments and evaluations come back for the transcript
 *
 * [DEP] halo2-axiom plonk/prover.rs create_proof, plonk/permutation/prover.rs, plonk/lookup/prover.rs, plonk/evaluation.rs -- reached from
 * /root/reference/aggregator/src/wrapper.rs:129 (gen_snark) and /root/reference/aggregator/benches/state_transition_circuit.rs:84 (k = 15);
 * shape: `gate_cols` advice columns with halo2-lib's vertical gate, one range-lookup column, one constants column
 * (/root/reference/aggregator/benches/wrapper_circuit.rs:41-48).  No Rust toolchain exists in the build image, so this is how the patch's
 * behaviour is exercised on the GPU: every commitment, every evaluation and the quotient's coefficients of sequences B and D must equal
 * those of sequence H (tests/test_gpu_prover_sequence.py builds and runs it; bench.py runs `--device-only` at the wrapper's k = 22).
 *
 * The row programs (the `GraphEvaluator` translation, the grand products' numerators / denominators) and the domain constants come from a
 * record written by zksnap_circuits_halo2_amd/evaluation.py export_prover_programs -- on the Rust side the patched evaluation.rs produces
 * the same zkhip_vm_program from its own graph.  Challenges are seeded there: no transcript on this side.  The witness is random (the
 * lookup column holds table values), so the constraint system is not satisfied and h is not the true quotient; all three sequences compute
 * the same h from the same columns, which is what is compared.  (Satisfied circuits: tools/prove_flow.py, tests/test_gpu_prover_flow.py.)
 *
 * usage: prover_sequence <programs.bin> [--device-only [repeats]]        exit status 0 = every comparison passed
 * (--device-only: sequence D alone, `repeats` times over the same buffers -- the first proof of a process runs at cold clocks -- reporting
 * the first and the fastest) */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zkhip.h"

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: %s (%s)\n", __LINE__, what, zkhip_last_error()); failures++; } } while (0)
#define OK(call) CHECK((call) == ZKHIP_OK, #call)

/* ---- the record -------------------------------------------------------------------------------------------------------------------- */
static const unsigned char *rd_ptr, *rd_end;
static const void *rd(size_t bytes) {
  const void *p = rd_ptr;
  if ((size_t)(rd_end - rd_ptr) < bytes) { fprintf(stderr, "record truncated\n"); exit(2); }
  rd_ptr += bytes;
  return p;
}
static uint32_t rd_u32(void) { uint32_t v; memcpy(&v, rd(4), 4); return v; }
static void *rd_copy(size_t bytes) {          /* an aligned copy of the next `bytes` of the record */
  void *p = malloc(bytes ? bytes : 1);
  if (!p) { fprintf(stderr, "out of host memory\n"); exit(2); }
  memcpy(p, rd(bytes), bytes);
  return p;
}
static void rd_fr(uint64_t out[4]) { memcpy(out, rd(32), 32); }
static void rd_prog(zkhip_vm_program *p) {
  uint32_t has_omega;
  int32_t scale;
  memset(p, 0, sizeof(*p));
  p->n_insns = rd_u32();
  p->insns = (const zkhip_vm_insn *)rd_copy((size_t)p->n_insns * sizeof(zkhip_vm_insn));
  p->n_constants = rd_u32();
  p->constants = (const uint64_t *)rd_copy((size_t)p->n_constants * 32);
  p->n_rotations = rd_u32();
  p->rotations = (const int32_t *)rd_copy((size_t)p->n_rotations * 4);
  memcpy(&scale, rd(4), 4);
  p->rot_scale = scale;
  p->result_reg = rd_u32();
  has_omega = rd_u32();
  p->omega = has_omega ? (const uint64_t *)rd_copy(32) : NULL;
}

/* ---- helpers ----------------------------------------------------------------------------------------------------------------------- */
static uint64_t rng_state = 0x5A4B534E41500007ULL;
static uint64_t next_u64(void) { rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27; return rng_state * 0x2545F4914F6CDD1DULL; }
static void fill_random(uint64_t *s, size_t n) {            /* canonical Montgomery word patterns below 2^252 */
  size_t i;
  for (i = 0; i < n; i++) { s[4 * i] = next_u64(); s[4 * i + 1] = next_u64(); s[4 * i + 2] = next_u64(); s[4 * i + 3] = next_u64() & 0x0fffffffffffffffULL; }
}
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e3 + (double)t.tv_nsec / 1e6; }
static void *xmalloc(size_t b) { void *p = malloc(b ? b : 1); if (!p) { fprintf(stderr, "out of host memory (%zu bytes)\n", b); exit(2); } return p; }
static void *dalloc(size_t b) { void *p = NULL; if (zkhip_alloc(b, &p) != ZKHIP_OK) { fprintf(stderr, "zkhip_alloc(%zu): %s\n", b, zkhip_last_error()); exit(2); } return p; }
static int same_points(const uint64_t *a_xyz, const uint64_t *b_xyz, size_t count) {
  uint64_t *in = xmalloc(count * 2 * 96), *out = xmalloc(count * 2 * 64);
  int ok;
  memcpy(in, a_xyz, count * 96); memcpy(in + 12 * count, b_xyz, count * 96);
  ok = zkhip_g1_batch_normalize(in, 2 * count, out) == ZKHIP_OK && memcmp(out, out + 8 * count, count * 64) == 0;
  free(in); free(out);
  return ok;
}

/* results of one sequence, for the comparison */
typedef struct {
  uint64_t *commits; size_t n_commits;       /* Jacobian, in the order they would enter the transcript */
  uint64_t *evals; size_t n_evals;
  uint64_t *h; size_t h_len;                 /* 3n coefficients of the quotient (NULL in --device-only runs at large k) */
  double ms;
} result_t;

int main(int argc, char **argv) {
  int device_only = argc > 2 && strcmp(argv[2], "--device-only") == 0;
  int repeats = device_only && argc > 3 && atoi(argv[3]) > 1 ? atoi(argv[3]) : 1, rep;
  double first_ms = 0, with_open_ms = 0;
  FILE *f;
  long fsz;
  unsigned char *blob;
  uint32_t k, ek, G, NL, nperm, nsets, chunk, blind, ncol, q_fixed, q_advice, q_l0, q_sigma, q_perm, q_lookup, period, i, j, si;
  uint64_t om_inv[4], divisor[4], ext_om[4], ext_om_inv[4], ext_div[4], zeta[4], xpt[4], omega[4], delta[4], beta[4], gamma[4];
  uint64_t rotpt[6][4], y_mo[4], v_mo[4], u_mo[4];     /* x omega^r for r = 0, 1, 2, 3, -1, usable; the multi-open argument's challenges */
  size_t mo_queries = 0, mo_witnesses = 0;
  int mo_refused = 0;
  const uint64_t *t_eval;
  zkhip_vm_program to_mont, *perm_num, *perm_den, lk_num, lk_den, eval_h, *eval_h_parts = NULL;
  uint32_t n_parts = 0;
  const uint64_t *part_weights = NULL;
  size_t n, en, u, nadv, nproof, first_proof2;
  uint64_t *g, *gl, **lag, *advice_flat;
  void *d_g, *d_gl;
  result_t R[3];
  if (argc < 2) { fprintf(stderr, "usage: %s programs.bin [--device-only]\n", argv[0]); return 2; }
  f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); fsz = ftell(f); fseek(f, 0, SEEK_SET);
  blob = xmalloc((size_t)fsz);
  if (fread(blob, 1, (size_t)fsz, f) != (size_t)fsz) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);
  rd_ptr = blob; rd_end = blob + fsz;
  if (memcmp(rd(4), "ZKPS", 4) != 0 || rd_u32() != 3) { fprintf(stderr, "not a prover-sequence record (version 3)\n"); return 2; }
  k = rd_u32(); ek = rd_u32(); G = rd_u32(); NL = rd_u32(); nperm = rd_u32(); nsets = rd_u32(); chunk = rd_u32(); blind = rd_u32();
  ncol = rd_u32(); q_fixed = rd_u32(); q_advice = rd_u32(); q_l0 = rd_u32(); q_sigma = rd_u32(); q_perm = rd_u32(); q_lookup = rd_u32();
  rd_fr(om_inv); rd_fr(divisor); rd_fr(ext_om); rd_fr(ext_om_inv); rd_fr(ext_div); rd_fr(zeta); rd_fr(xpt); rd_fr(omega); rd_fr(delta); rd_fr(beta); rd_fr(gamma);
  for (i = 0; i < 6; i++) rd_fr(rotpt[i]);
  rd_fr(y_mo); rd_fr(v_mo); rd_fr(u_mo);
  period = rd_u32();
  t_eval = (const uint64_t *)rd_copy((size_t)period * 32);
  if (NL != 1 || q_fixed != 0) { fprintf(stderr, "shape not supported by this program\n"); return 2; }
  perm_num = xmalloc((size_t)(nsets ? nsets : 1) * sizeof(*perm_num)); perm_den = xmalloc((size_t)(nsets ? nsets : 1) * sizeof(*perm_den));
  rd_prog(&to_mont);
  for (si = 0; si < nsets; si++) rd_prog(&perm_num[si]);
  for (si = 0; si < nsets; si++) rd_prog(&perm_den[si]);
  rd_prog(&lk_num); rd_prog(&lk_den); rd_prog(&eval_h);
  n_parts = rd_u32();            /* the quotient numerator as a sum of programs (long program, few rows): count, weights, programs */
  if (n_parts) {
    part_weights = (const uint64_t *)rd_copy((size_t)n_parts * 32);
    eval_h_parts = xmalloc((size_t)n_parts * sizeof(*eval_h_parts));
    for (i = 0; i < n_parts; i++) rd_prog(&eval_h_parts[i]);
  }
  n = (size_t)1 << k; en = (size_t)1 << ek; u = n - (blind + 1);
  nadv = G + NL;
  /* witness-dependent columns of the quotient, in column order: advice [q_advice, q_l0) and products / permuted pair [q_perm, ncol) */
  nproof = nadv + (ncol - q_perm);
  first_proof2 = q_perm;
  printf("shape: k = %u, extended k = %u, %u gate columns + %u lookup, %u permutation columns in %u sets, %u quotient columns (%zu per proof)\n",
         k, ek, G, NL, nperm, nsets, ncol, nproof);

  OK(zkhip_init(NULL, 0));

  /* ---- ParamsKZG::setup (commitment_patch.rs): g, g_lagrange = g_to_lagrange(g), both pinned ---- */
  g = xmalloc(n * 64); gl = xmalloc(n * 64);
  {
    uint64_t t0[4] = {5, 0, 0, 0}, d[4] = {7, 0, 0, 0};
    d_g = dalloc(n * 64); d_gl = dalloc(n * 64);
    OK(zkhip_g1_gen_walk_device(t0, d, n, d_g, NULL));
    OK(zkhip_g_to_lagrange_device(d_g, k, d_gl, NULL));
    OK(zkhip_sync());
    OK(zkhip_download(g, d_g, n * 64)); OK(zkhip_download(gl, d_gl, n * 64));
    OK(zkhip_free(d_g)); OK(zkhip_free(d_gl));
    OK(zkhip_register_bases(g, n)); OK(zkhip_register_bases(gl, n));
  }

  /* ---- the circuit's columns in the Lagrange basis (host): proving-key columns + the witness ---- */
  lag = xmalloc(ncol * sizeof(*lag));
  for (i = 0; i < ncol; i++) lag[i] = NULL;
  {
    /* fixed: G selectors (1 on gate rows), the constants column (random), the table (row mod 2^bits as field elements) */
    const uint32_t bits = k - 1 < 8 ? k - 1 : 8;
    uint64_t one[4], *raw = xmalloc(n * 32);
    const uint64_t *cols1[1];
    size_t r;
    memset(raw, 0, n * 32); raw[0] = 1;
    cols1[0] = raw;
    { uint64_t *tmp = xmalloc(n * 32); OK(zkhip_fr_eval_rows(&to_mont, cols1, 1, k, 0, tmp)); memcpy(one, tmp, 32); free(tmp); }
    for (i = 0; i < G; i++) {
      lag[q_fixed + i] = xmalloc(n * 32);
      memset(lag[q_fixed + i], 0, n * 32);
      for (r = 0; r + 3 < u; r += 4) memcpy(lag[q_fixed + i] + 4 * r, one, 32);
    }
    lag[q_fixed + G] = xmalloc(n * 32); fill_random(lag[q_fixed + G], n);
    for (r = 0; r < n; r++) { raw[4 * r] = r & ((1u << bits) - 1); raw[4 * r + 1] = raw[4 * r + 2] = raw[4 * r + 3] = 0; }
    lag[q_fixed + G + 1] = xmalloc(n * 32);
    OK(zkhip_fr_eval_rows(&to_mont, cols1, 1, k, 0, lag[q_fixed + G + 1]));
    /* advice: gate columns random, the lookup column = table values in random order (rows >= u: blinding) */
    advice_flat = xmalloc(nadv * n * 32);
    for (i = 0; i < G; i++) fill_random(advice_flat + (size_t)i * n * 4, n);
    for (r = 0; r < n; r++) { raw[4 * r] = next_u64() & ((1u << bits) - 1); }
    OK(zkhip_fr_eval_rows(&to_mont, cols1, 1, k, 0, advice_flat + (size_t)G * n * 4));
    fill_random(advice_flat + ((size_t)G * n + u) * 4, n - u);
    for (i = 0; i < nadv; i++) lag[q_advice + i] = advice_flat + (size_t)i * n * 4;
    /* l_0, l_last, l_active_row */
    for (i = 0; i < 3; i++) { lag[q_l0 + i] = xmalloc(n * 32); memset(lag[q_l0 + i], 0, n * 32); }
    memcpy(lag[q_l0], one, 32);
    memcpy(lag[q_l0 + 1] + 4 * u, one, 32);
    for (r = 0; r < u; r++) memcpy(lag[q_l0 + 2] + 4 * r, one, 32);
    /* sigma columns: random field elements (a real key holds delta^i omega^j patterns; any column exercises the same calls) */
    for (i = 0; i < nperm; i++) { lag[q_sigma + i] = xmalloc(n * 32); fill_random(lag[q_sigma + i], n); }
    free(raw);
  }
  /* permutation column c: advice c for c < nadv, the constants column after them */
#define PCOL(c) ((c) < nadv ? lag[q_advice + (c)] : lag[q_fixed + G])

  /* blinding rows of the products and of the permuted pair, shared by the three sequences (the host's RNG in create_proof) */
  {
    uint64_t *blind_rows = xmalloc((size_t)(nsets + 3) * (n - u) * 32);
    fill_random(blind_rows, (size_t)(nsets + 3) * (n - u));
#define BLIND_ROWS(slot) (blind_rows + (size_t)(slot) * (n - u) * 4)

    /* ================================================= sequences H and B (host buffers) ================================================= */
    int mode;
    for (mode = 0; mode < 2 && !device_only; mode++) {
      result_t *res = &R[mode];
      uint64_t **col = xmalloc(ncol * sizeof(*col));       /* this sequence's working copies of the witness-dependent columns (Lagrange, then coefficients) */
      uint64_t **ext = xmalloc(ncol * sizeof(*ext));
      uint64_t *h_ext, *hc, *proof_flat = NULL, *ext_flat = NULL;
      const uint64_t **cptr = xmalloc(ncol * 2 * sizeof(*cptr));
      size_t c = 0, e = 0, p;
      double t0;
      res->n_commits = nadv + 2 + nsets + 1 + 3;
      res->commits = xmalloc(res->n_commits * 96);
      res->n_evals = nproof + 3;
      res->evals = xmalloc(res->n_evals * 32);
      /* proving-key columns: coefficient form and extended cosets are part of the key (keygen, outside the proof) */
      for (i = 0; i < ncol; i++) { col[i] = NULL; ext[i] = NULL; }
      for (i = 0; i < ncol; i++) {
        if ((i >= q_advice && i < q_l0) || i >= q_perm) continue;
        col[i] = xmalloc(n * 32); ext[i] = xmalloc(en * 32);
        memcpy(col[i], lag[i], n * 32);
        OK(zkhip_ifft_scaled(col[i], om_inv, k, divisor));
        OK(zkhip_coeff_to_extended(col[i], k, ext[i], ek, ext_om, zeta));
      }
      if (mode == 1) { proof_flat = xmalloc(nproof * n * 32); ext_flat = xmalloc(nproof * en * 32); }
      for (p = 0; p < nproof; p++) {
        const size_t ci = p < nadv ? q_advice + p : first_proof2 + (p - nadv);
        col[ci] = mode == 1 ? proof_flat + p * n * 4 : xmalloc(n * 32);
        ext[ci] = mode == 1 ? ext_flat + p * en * 4 : xmalloc(en * 32);
      }
      t0 = now_ms();
      /* 1. advice commitments (Lagrange basis) */
      for (i = 0; i < nadv; i++) memcpy(col[q_advice + i], lag[q_advice + i], n * 32);
      if (mode == 0) for (i = 0; i < nadv; i++) OK(zkhip_msm_g1(col[q_advice + i], gl, n, res->commits + 12 * c++));
      else { OK(zkhip_msm_g1_batch(col[q_advice], gl, n, nadv, res->commits + 12 * c)); c += nadv; }
      /* 2. lookup: permuted input / table (+ blinding rows), commitments */
      {
        uint64_t *pa = col[q_lookup + 1], *ps = col[q_lookup + 2];
        OK(zkhip_lookup_permute(lag[q_advice + G], lag[q_fixed + G + 1], u, pa, ps));
        memcpy(pa + 4 * u, BLIND_ROWS(nsets + 1), (n - u) * 32);
        memcpy(ps + 4 * u, BLIND_ROWS(nsets + 2), (n - u) * 32);
        if (mode == 0) { OK(zkhip_msm_g1(pa, gl, n, res->commits + 12 * c++)); OK(zkhip_msm_g1(ps, gl, n, res->commits + 12 * c++)); }
        else { OK(zkhip_msm_g1_batch(pa, gl, n, 2, res->commits + 12 * c)); c += 2; }        /* permuted_input and permuted_table are adjacent */
      }
      /* 3. grand products: permutation sets (chained), lookup; commitments */
      {
        uint64_t *den = xmalloc(n * 32), last[4] = {0, 0, 0, 0};
        if (mode == 1) {                                  /* every set in ONE call; z sets are adjacent in proof_flat: the dense [sets][n] layout */
          for (j = 0; j < nperm; j++) { cptr[j] = PCOL(j); cptr[nperm + j] = lag[q_sigma + j]; }
          OK(zkhip_permutation_products(cptr, cptr + nperm, nperm, chunk, k, u, beta, gamma, delta, omega, col[q_perm]));
          for (si = 0; si < nsets; si++) memcpy(col[q_perm + si] + 4 * (u + 1), BLIND_ROWS(si) + 4, (n - u - 1) * 32);
        }
        for (si = 0; si < nsets && mode == 0; si++) {
          const uint32_t lo = si * chunk, hi = lo + chunk < nperm ? lo + chunk : nperm, cnt = hi - lo;
          uint64_t *z = col[q_perm + si];
          for (j = 0; j < cnt; j++) { cptr[j] = PCOL(lo + j); cptr[cnt + j] = lag[q_sigma + lo + j]; }
          OK(zkhip_fr_eval_rows(&perm_num[si], cptr, cnt, k, 0, z));
          OK(zkhip_fr_eval_rows(&perm_den[si], cptr, 2 * cnt, k, 0, den));
          OK(zkhip_fr_grand_product(z, den, n, z));
          if (si > 0) {                                   /* z_k[0] = z_{k-1}[u]: scale by the previous set's last usable value */
            zkhip_vm_insn ins;
            zkhip_vm_program sc;
            const int32_t rot0 = 0;
            memset(&ins, 0, sizeof(ins)); memset(&sc, 0, sizeof(sc));
            ins.op = ZKHIP_OP_MUL; ins.dst = 0; ins.a.kind = ZKHIP_SRC_COLUMN; ins.a.index = 0; ins.a.rot = 0; ins.b.kind = ZKHIP_SRC_CONST; ins.b.index = 0;
            sc.insns = &ins; sc.n_insns = 1; sc.constants = last; sc.n_constants = 1; sc.rotations = &rot0; sc.n_rotations = 1; sc.rot_scale = 1; sc.result_reg = 0;
            memcpy(den, z, n * 32);                       /* (den is free again: read the copy, write z) */
            cptr[0] = den;
            OK(zkhip_fr_eval_rows(&sc, cptr, 1, k, 0, z));
          }
          memcpy(last, z + 4 * u, 32);
          memcpy(z + 4 * (u + 1), BLIND_ROWS(si) + 4, (n - u - 1) * 32);
        }
        {
          uint64_t *zl = col[q_lookup];
          cptr[0] = lag[q_advice + G]; cptr[1] = lag[q_fixed + G + 1];
          OK(zkhip_fr_eval_rows(&lk_num, cptr, 2, k, 0, zl));
          cptr[0] = col[q_lookup + 1]; cptr[1] = col[q_lookup + 2];
          OK(zkhip_fr_eval_rows(&lk_den, cptr, 2, k, 0, den));
          OK(zkhip_fr_grand_product(zl, den, n, zl));
          memcpy(zl + 4 * (u + 1), BLIND_ROWS(nsets) + 4, (n - u - 1) * 32);
        }
        free(den);
        if (mode == 0) for (i = 0; i < nsets + 1; i++) OK(zkhip_msm_g1(col[q_perm + i], gl, n, res->commits + 12 * c++));
        else { OK(zkhip_msm_g1_batch(col[q_perm], gl, n, nsets + 1, res->commits + 12 * c)); c += nsets + 1; }   /* z sets and the lookup product are adjacent */
      }
      /* 4. lagrange_to_coeff, 5. coeff_to_extended of the witness-dependent columns */
      if (mode == 0) {
        for (p = 0; p < nproof; p++) {
          const size_t ci = p < nadv ? q_advice + p : first_proof2 + (p - nadv);
          OK(zkhip_ifft_scaled(col[ci], om_inv, k, divisor));
          OK(zkhip_coeff_to_extended(col[ci], k, ext[ci], ek, ext_om, zeta));
        }
      } else {
        OK(zkhip_ifft_scaled_batch(proof_flat, om_inv, k, divisor, (uint32_t)nproof));
        OK(zkhip_coeff_to_extended_batch(proof_flat, k, ext_flat, ek, (uint32_t)nproof, ext_om, zeta));
      }
      /* 6. quotient: row program over the extended coset, / (X^n - 1), back to coefficients, commitments of the three pieces */
      h_ext = xmalloc(en * 32); hc = xmalloc(3 * n * 32);
      for (i = 0; i < ncol; i++) cptr[i] = ext[i];
      OK(zkhip_fr_eval_rows(&eval_h, cptr, ncol, ek, 0, h_ext));
      OK(zkhip_mul_periodic(h_ext, en, t_eval, period));
      OK(zkhip_extended_to_coeff(h_ext, ek, ext_om_inv, ext_div, zeta, hc, 3 * n));
      if (mode == 0) for (i = 0; i < 3; i++) OK(zkhip_msm_g1(hc + (size_t)i * n * 4, g, n, res->commits + 12 * c++));
      else { OK(zkhip_msm_g1_batch(hc, g, n, 3, res->commits + 12 * c)); c += 3; }
      /* 7. evaluations at x */
      for (p = 0; p < nproof; p++) {
        const size_t ci = p < nadv ? q_advice + p : first_proof2 + (p - nadv);
        OK(zkhip_fr_eval_polynomial(col[ci], n, xpt, res->evals + 4 * e++));
      }
      for (i = 0; i < 3; i++) OK(zkhip_fr_eval_polynomial(hc + (size_t)i * n * 4, n, xpt, res->evals + 4 * e++));
      res->ms = now_ms() - t0;
      CHECK(c == res->n_commits && e == res->n_evals, "commitment / evaluation count");
      res->h = hc; res->h_len = 3 * n;
      free(h_ext);
      for (i = 0; i < ncol; i++) {
        const int proof_col = (i >= q_advice && i < q_l0) || i >= q_perm;
        if (!proof_col || mode == 0) { free(col[i]); free(ext[i]); }
      }
      free(proof_flat); free(ext_flat); free(col); free(ext); free((void *)cptr);
    }

    /* ========================================================= sequence D (device-resident) ========================================================= */
    {
      result_t *res = &R[2];
      char *d_lag, *d_coeff, *d_ext, *d_tmp, *d_den, *d_hext, *d_hc, *d_out, *d_teval, *d_table_lag, *d_sigma_lag, *d_const_lag;
      const void **dptr = xmalloc(ncol * 2 * sizeof(*dptr));
      size_t c = 0, p;
      double t0;
      res->ms = 0;
      res->n_commits = nadv + 2 + nsets + 1 + 3;
      res->commits = xmalloc(res->n_commits * 96);
      res->n_evals = nproof + 3;
      res->evals = xmalloc(res->n_evals * 32);
      /* device memory: all quotient columns in coefficient form [ncol][n] and on the extended coset [ncol][en]; the proving key's
       * part is uploaded / transformed once (keygen), the witness-dependent part per proof */
      d_coeff = dalloc((size_t)ncol * n * 32); d_ext = dalloc((size_t)ncol * en * 32);
      d_lag = dalloc(nproof * n * 32);                     /* Lagrange-basis working columns of the proof: advice | z sets | z_lookup | a' | s' */
      d_tmp = dalloc(n * 32); d_den = dalloc(n * 32); d_hext = dalloc(en * 32); d_hc = dalloc(en * 32);
      d_out = dalloc((res->n_commits * 96 + res->n_evals * 32 + 255) & ~(size_t)255);
      d_teval = dalloc((size_t)period * 32);
      d_table_lag = dalloc(n * 32); d_sigma_lag = dalloc((size_t)nperm * n * 32); d_const_lag = dalloc(n * 32);
      OK(zkhip_upload(d_teval, t_eval, (size_t)period * 32));
      OK(zkhip_upload(d_table_lag, lag[q_fixed + G + 1], n * 32));
      OK(zkhip_upload(d_const_lag, lag[q_fixed + G], n * 32));
      for (i = 0; i < nperm; i++) OK(zkhip_upload(d_sigma_lag + (size_t)i * n * 32, lag[q_sigma + i], n * 32));
      for (i = 0; i < ncol; i++) {                          /* keygen: the key's columns, coefficient form and coset, in HBM */
        if ((i >= q_advice && i < q_l0) || i >= q_perm) continue;
        OK(zkhip_upload(d_coeff + (size_t)i * n * 32, lag[i], n * 32));
        OK(zkhip_ifft_scaled_device(d_coeff + (size_t)i * n * 32, om_inv, k, divisor, NULL));
        OK(zkhip_coeff_to_extended_device(d_coeff + (size_t)i * n * 32, n, k, d_ext + (size_t)i * en * 32, en, ek, 1, ext_om, zeta, NULL));
      }
      OK(zkhip_sync());
#define DLAG(p_) (d_lag + (size_t)(p_) * n * 32)            /* proof column p_ (0 .. nproof): advice, then z sets, z_lookup, a', s' */
#define DPCOL(c_) ((c_) < nadv ? (const void *)DLAG(c_) : (const void *)d_const_lag)
      for (rep = 0; rep < repeats; rep++) {
      c = 0;
      t0 = now_ms();
      /* 1. the witness goes up once; advice commitments in ONE call against the pinned g_lagrange */
      OK(zkhip_upload(d_lag, advice_flat, nadv * n * 32));
      OK(zkhip_msm_g1_registered_batch_device(gl, d_lag, n, nadv, n, d_out + 96 * c, NULL)); c += nadv;
      /* 2. lookup: permuted pair on the device, blinding rows from the host's RNG, one commit call */
      {
        char *pa = DLAG(nadv + nsets + 1), *ps = DLAG(nadv + nsets + 2);
        OK(zkhip_lookup_permute_device(DLAG(G), d_table_lag, u, pa, ps, NULL));
        OK(zkhip_upload(pa + u * 32, BLIND_ROWS(nsets + 1), (n - u) * 32));
        OK(zkhip_upload(ps + u * 32, BLIND_ROWS(nsets + 2), (n - u) * 32));
        OK(zkhip_msm_g1_registered_batch_device(gl, pa, n, 2, n, d_out + 96 * c, NULL)); c += 2;
      }
      /* 3. grand products on the device: every permutation set in ONE call (chained on the device: no read-back), then the lookup's */
      {
        for (j = 0; j < nperm; j++) { dptr[j] = DPCOL(j); dptr[nperm + j] = d_sigma_lag + (size_t)j * n * 32; }
        OK(zkhip_permutation_products_device(dptr, dptr + nperm, nperm, chunk, k, u, beta, gamma, delta, omega, DLAG(nadv), NULL));
        for (si = 0; si < nsets; si++) OK(zkhip_upload(DLAG(nadv + si) + (u + 1) * 32, BLIND_ROWS(si) + 4, (n - u - 1) * 32));
        {
          char *zl = DLAG(nadv + nsets);
          dptr[0] = DLAG(G); dptr[1] = d_table_lag;
          OK(zkhip_fr_eval_rows_device(&lk_num, dptr, 2, k, 0, zl, NULL));
          dptr[0] = DLAG(nadv + nsets + 1); dptr[1] = DLAG(nadv + nsets + 2);
          OK(zkhip_fr_eval_rows_device(&lk_den, dptr, 2, k, 0, d_den, NULL));
          OK(zkhip_fr_grand_product_device(zl, d_den, n, zl, NULL));
          OK(zkhip_upload(zl + (u + 1) * 32, BLIND_ROWS(nsets) + 4, (n - u - 1) * 32));
        }
        OK(zkhip_msm_g1_registered_batch_device(gl, DLAG(nadv), n, nsets + 1, n, d_out + 96 * c, NULL)); c += nsets + 1;
      }
      /* 4. lagrange_to_coeff of every witness-dependent column in ONE call (in place), then into the quotient's column order */
      OK(zkhip_ifft_scaled_batch_device(d_lag, om_inv, k, divisor, (uint32_t)nproof, n, NULL));
      /* 5. coeff_to_extended: advice block and product block, one call each, straight into their slots of the coset array */
      OK(zkhip_coeff_to_extended_device(DLAG(0), n, k, d_ext + (size_t)q_advice * en * 32, en, ek, (uint32_t)nadv, ext_om, zeta, NULL));
      OK(zkhip_coeff_to_extended_device(DLAG(nadv), n, k, d_ext + (size_t)q_perm * en * 32, en, ek, (uint32_t)(nproof - nadv), ext_om, zeta, NULL));
      /* 6. quotient */
      for (i = 0; i < ncol; i++) dptr[i] = d_ext + (size_t)i * en * 32;
      if (n_parts) OK(zkhip_fr_eval_rows_sum_device(eval_h_parts, part_weights, n_parts, dptr, ncol, ek, d_hext, NULL));   /* the parts side by side */
      else OK(zkhip_fr_eval_rows_device(&eval_h, dptr, ncol, ek, 0, d_hext, NULL));
      OK(zkhip_mul_periodic_device(d_hext, en, d_teval, period, NULL));
      OK(zkhip_extended_to_coeff_device(d_hext, en, ek, ext_om_inv, ext_div, zeta, d_hc, en, 3 * n, 1, NULL));
      OK(zkhip_msm_g1_registered_batch_device(g, d_hc, n, 3, n, d_out + 96 * c, NULL)); c += 3;
      /* 7. evaluations at x: one call for every opened polynomial */
      for (p = 0; p < nproof; p++) dptr[p] = DLAG(p);
      for (i = 0; i < 3; i++) dptr[nproof + i] = d_hc + (size_t)i * n * 32;
      OK(zkhip_fr_eval_polynomial_batch_device(dptr, nproof + 3, n, xpt, d_out + 96 * res->n_commits, NULL));
      /* the transcript's view: commitments and evaluations, one read-back */
      OK(zkhip_download(res->commits, d_out, res->n_commits * 96));
      OK(zkhip_download(res->evals, d_out + 96 * res->n_commits, res->n_evals * 32));
      {
        const double ms = now_ms() - t0;                   /* phases 1-7: what sequences H and B time as well */
        if (rep == 0) first_ms = ms;
        if (rep == 0 || ms < res->ms) res->ms = ms;
      }
      /* 8. multi-open argument on the device-resident polynomials (prover_patch.rs `open_on_device`): the query plan of create_proof for this
       * shape -- fixed at x; gate advice at x, omega x, omega^2 x, omega^3 x; lookup advice and the sigma polynomials at x; every permutation
       * product at x and omega x, all but the last also at omega^u x; the lookup's product at x and omega x, its permuted input at x and
       * omega^-1 x, its permuted table at x; the quotient's pieces at x -- through SHPLONK, the benches' argument
       * (/root/reference/aggregator/benches/wrapper_circuit.rs:140); the library evaluates (has_eval = 0) */
      {
        const size_t max_q = (size_t)(G + 2) + 4 * (size_t)G + 1 + nperm + 3 * (size_t)nsets + 5 + 3;
        zkhip_prover_query *q = xmalloc(max_q * sizeof(*q));
        zkhip_shplonk *st = NULL;
        uint64_t hh[12], hp[12];
        size_t nq = 0;
#define ADDQ(ptr_, rot_) do { memset(&q[nq], 0, sizeof(q[nq])); memcpy(q[nq].point, rotpt[rot_], 32); q[nq].d_poly = (ptr_); nq++; } while (0)
#define DKEY(col_) (d_coeff + (size_t)(col_) * n * 32)       /* a proving-key column in coefficient form */
        for (i = q_fixed; i < q_advice; i++) ADDQ(DKEY(i), 0);
        for (i = 0; i < G; i++) for (j = 0; j < 4; j++) ADDQ(DLAG(i), j);
        ADDQ(DLAG(G), 0);
        for (i = 0; i < nperm; i++) ADDQ(DKEY(q_sigma + i), 0);
        for (si = 0; si < nsets; si++) { ADDQ(DLAG(nadv + si), 0); ADDQ(DLAG(nadv + si), 1); if (si + 1 < nsets) ADDQ(DLAG(nadv + si), 5); }
        ADDQ(DLAG(nadv + nsets), 0); ADDQ(DLAG(nadv + nsets), 1);
        ADDQ(DLAG(nadv + nsets + 1), 0); ADDQ(DLAG(nadv + nsets + 1), 4);
        ADDQ(DLAG(nadv + nsets + 2), 0);
        for (i = 0; i < 3; i++) ADDQ(d_hc + (size_t)i * n * 32, 0);
        CHECK(nq <= max_q, "query count");
        mo_queries = nq;
        OK(zkhip_multiopen_shplonk_begin_device(g, k, q, nq, y_mo, v_mo, hh, &st));
        OK(zkhip_multiopen_shplonk_finish_device(st, u_mo, hp));
        {
          const double ms = now_ms() - t0;                 /* the whole device side of the proof */
          if (rep == 0 || ms < with_open_ms) with_open_ms = ms;
        }
        if (!device_only) {                                /* outside the timing: GWC (the gen_snark path's argument) on the same queries; a wrong evaluation */
          uint64_t *wit = xmalloc(8 * 96), aff[16];
          OK(zkhip_multiopen_gwc_device(g, k, q, nq, v_mo, wit, 8, &mo_witnesses));
          memcpy(wit, hh, 96); memcpy(wit + 12, hp, 96);
          CHECK(zkhip_g1_batch_normalize(wit, 2, aff) == ZKHIP_OK && (aff[0] | aff[1] | aff[2] | aff[3]) != 0 && (aff[8] | aff[9] | aff[10] | aff[11]) != 0,
                "H and H' are points other than the identity");
          memcpy(q[1].eval, res->evals, 32);               /* some other polynomial's evaluation: finish must refuse (L(u) != 0) and free the state */
          q[1].has_eval = 1;
          OK(zkhip_multiopen_shplonk_begin_device(g, k, q, nq, y_mo, v_mo, hh, &st));
          mo_refused = zkhip_multiopen_shplonk_finish_device(st, u_mo, hp) == ZKHIP_EINVAL;
          free(wit);
        }
        free(q);
      }
      }
      CHECK(c == res->n_commits, "commitment count");
      res->h = NULL; res->h_len = 0;
      if (!device_only) { res->h = xmalloc(3 * n * 32); res->h_len = 3 * n; OK(zkhip_download(res->h, d_hc, 3 * n * 32)); }
      (void)d_tmp;
      OK(zkhip_free(d_lag)); OK(zkhip_free(d_coeff)); OK(zkhip_free(d_ext)); OK(zkhip_free(d_tmp)); OK(zkhip_free(d_den)); OK(zkhip_free(d_hext));
      OK(zkhip_free(d_hc)); OK(zkhip_free(d_out)); OK(zkhip_free(d_teval)); OK(zkhip_free(d_table_lag)); OK(zkhip_free(d_sigma_lag)); OK(zkhip_free(d_const_lag));
      free((void *)dptr);
    }
    free(blind_rows);
  }

  /* ---- comparison ---- */
  if (!device_only) {
    int m;
    for (m = 1; m < 3; m++) {
      const char *name = m == 1 ? "B (one call per phase)" : "D (device-resident)";
      CHECK(R[m].n_commits == R[0].n_commits && same_points(R[m].commits, R[0].commits, R[0].n_commits), name);
      CHECK(R[m].n_evals == R[0].n_evals && memcmp(R[m].evals, R[0].evals, R[0].n_evals * 32) == 0, name);
      CHECK(R[m].h_len == R[0].h_len && memcmp(R[m].h, R[0].h, R[0].h_len * 32) == 0, name);
    }
    {
      size_t nz = 0, r;
      for (r = 0; r < R[0].h_len * 4; r++) nz += R[0].h[r] != 0;
      CHECK(nz > R[0].h_len, "the quotient's coefficients are not trivially zero");
    }
    printf("sequence_ms host_call_by_call=%.2f host_one_call_per_phase=%.2f device_resident=%.2f (with the multi-open argument %.2f)\n", R[0].ms, R[1].ms, R[2].ms, with_open_ms);
    printf("commitments compared: %zu, evaluations compared: %zu, quotient coefficients compared: %zu\n", R[0].n_commits, R[0].n_evals, R[0].h_len);
    CHECK(mo_witnesses == 6 && mo_refused, "multi-open argument");
    printf("multiopen on the device: %zu queries, SHPLONK H and H', GWC %zu witnesses, a wrong evaluation %s\n", mo_queries, mo_witnesses, mo_refused ? "refused" : "ACCEPTED");
  } else {
    uint64_t aff[8];
    CHECK(zkhip_g1_batch_normalize(R[2].commits, 1, aff) == ZKHIP_OK, "first commitment normalises");
    printf("sequence_ms device_resident=%.2f first=%.2f repeats=%d with_multiopen=%.2f\n", R[2].ms, first_ms, repeats, with_open_ms);
  }
  OK(zkhip_unregister_bases(g)); OK(zkhip_unregister_bases(gl));
  zkhip_shutdown();
  if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
  printf("prover sequence OK\n");
  return 0;
}
