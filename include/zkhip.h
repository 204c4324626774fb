/* libzkhip -- MI355X (gfx950) kernels behind the halo2 prover's MSM / NTT boundary.  C ABI.
 *
 * Each entry point replaces one function of the un-vendored halo2-axiom crate [DEP] that the reference
 * reaches through `create_proof` (/root/reference/aggregator/src/wrapper.rs:129), `keygen_vk`/`keygen_pk`
 * (wrapper.rs:107-108) and `halo2_base::utils::testing::gen_proof`
 * (/root/reference/aggregator/benches/wrapper_circuit.rs:140).  SURVEY.md section 8(b) is the contract.
 *
 * Memory formats are exactly the Rust in-memory formats, so a Rust host passes slices through unchanged:
 *   Fr / Fq   : 4 x u64 little-endian limbs, Montgomery form (x * 2^256 mod p)            32 bytes
 *   G1Affine  : x || y  (identity = all-zero)                                             64 bytes
 *   G1        : x || y || z Jacobian (identity z = 0)                                     96 bytes
 * All buffers are caller-owned and borrowed for the duration of the call (registered bases: until
 * unregistered).  Every function returns ZKHIP_OK (0) or a negative ZKHIP_E* code and never aborts;
 * zkhip_last_error() describes the last failure on the calling thread.  There is no CPU fallback: without a
 * usable HIP device every compute entry point fails with ZKHIP_ENODEV.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_OK 0
#define ZKHIP_EINVAL (-1)  /* bad argument */
#define ZKHIP_ENODEV (-2)  /* no HIP device / init failed */
#define ZKHIP_EHIP (-3)    /* HIP runtime error, see zkhip_last_error() */
#define ZKHIP_ENOMEM (-4)  /* device allocation failed */
#define ZKHIP_EBUSY (-5)   /* zkhip_init with a different device list while host-buffer calls are running */

/* ---- lifecycle ----------------------------------------------------------------------------------- */
/* Name the HIP devices this process drives (`zkhip_init(const int *devices, int ndev)`).  devices == NULL
 * or ndev == 0: device 0 (or $ZKHIP_DEVICE).  devices[0] is the PRIMARY device: every `_device` entry point, every single transform and
 * every other vector operation runs there, and `_device` pointers are pointers into its memory.  With ndev > 1 the MSM over registered
 * bases is sharded by point range over all the devices (zkhip_register_bases below) and the polynomials of a BATCHED transform are
 * spread over them, each transform on one device (zkhip_set_ntt_fanout below) -- the 8 GPUs of one node behind one `create_proof`
 * process (/root/reference/aggregator/src/wrapper.rs:129).  Lazy init on first use is allowed (device 0).
 * Calling it again with the same list is a no-op; with a different list it shuts the library down first (ZKHIP_EBUSY while
 * host-buffer calls of other threads are still running).
 * Environment: ZKHIP_DEVICE (default device), ZKHIP_SHARDS (see zkhip_set_msm_shards), ZKHIP_HOST_LANES (1..4, default 2: host-buffer
 * calls that may be in flight at once, each with its own stream and scratch memory). */
int zkhip_init(const int *devices, int ndev);
void zkhip_shutdown(void);
const char *zkhip_last_error(void);
/* library / device identification, for logs: writes a NUL-terminated string (the primary device) */
int zkhip_device_name(char *buf, size_t len);
/* number of devices the library drives (0 when it cannot initialise) */
int zkhip_device_count(void);
/* Number of point-range shards zkhip_register_bases cuts an array into from now on: shard s of S covers points
 * [s n / S, (s + 1) n / S) (the first n % S shards one point more) and lives on device s % ndev.  Default: one shard per device.
 * More shards than devices ("virtual shards", also $ZKHIP_SHARDS) run the whole multi-GPU path -- per-shard tables, per-shard
 * Pippenger, gather of the 96-byte Jacobian partials, fold -- on fewer devices; the result is the same group element for every
 * shard count.  0 restores the default. */
int zkhip_set_msm_shards(int shards);
int zkhip_msm_shards(void);

/* Threads and streams.  Every entry point may be called from any host thread.  Host-buffer calls (no `_device` suffix) borrow one of
 * the library's lanes for the duration of the call, so up to ZKHIP_HOST_LANES of them run concurrently (one call's PCIe transfer
 * under another's kernels); further callers wait.  `_device` calls are asynchronous on the caller's stream and use scratch memory
 * that belongs to that stream: calls on different streams never share scratch, calls on one stream are ordered by the stream.  Two
 * host threads must not enqueue on the SAME stream at the same time (as with any HIP stream).  The profiling hooks
 * (zkhip_profile_*) are a single-caller debugging aid. */

/* ---- MSM: replaces `best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> G1` ------------------------- */
/* [DEP] halo2_proofs/src/arithmetic.rs; called by ParamsKZG::commit / commit_lagrange.  n may be any value
 * (n == 0 gives the identity).  out_xyz: Jacobian, canonical Montgomery limbs, any valid representative. */
int zkhip_msm_g1(const uint64_t *scalars, const uint64_t *bases, size_t n, uint64_t out_xyz[12]);

/* Multi-column commit: `batch` scalar vectors (contiguous, n elements each) against the same bases, e.g. every advice column of
 * a circuit; out_xyz: batch Jacobian points.  With registered bases this is one launch set (see the _device variant). */
int zkhip_msm_g1_batch(const uint64_t *scalars, const uint64_t *bases, size_t n, size_t batch, uint64_t *out_xyz);

/* The same over G2 (`best_multiexp::<G2Affine>`): bases n x 16 limbs (G2Affine = x.c0 || x.c1 || y.c0 || y.c1, Montgomery Fq, all-zero =
 * identity), out 24 limbs (G2 Jacobian x || y || z, each an Fq2).  No call site in the reference prover multiplies in G2 (it reads
 * params.g2() / params.s_g2(): /root/reference/aggregator/src/wrapper.rs:1142-1144); provided because the MSM is generic over the
 * curve.  General path only (per-window bucket sets + window fold). */
int zkhip_msm_g2(const uint64_t *scalars, const uint64_t *bases, size_t n, uint64_t out_xyz[24]);

/* Residency for `ParamsKZG::{g, g_lagrange}` (static per params object): upload once and build the prepared table
 * (2^(c w) * P_i for every window w: W * 64 bytes per point of HBM, one-time ~25 ms per 2^20 points; arrays of at most 2^15 points
 * -- $ZKHIP_DIRECT_MAX_LOG -- also get every multiple of every 8-bit window point, 256 KiB per point, and their single MSMs need no
 * buckets: DESIGN.md section 3c) on every shard's device;
 * zkhip_msm_g1 recognises `bases` pointers inside a registered range (any sub-range), skips the upload and runs the prepared
 * path: one shared bucket set per shard, no window fold, one gather + fold of the shards' partial sums.
 * Contract: the caller keeps [bases, bases + 8 n) alive AND UNCHANGED until zkhip_unregister_bases -- the table is built from the
 * contents at registration.  Recognition is by address, so memory that was freed without unregistering and later reused for other
 * points would alias a stale table; as a guard the library keeps 8 sampled points of the registered array and treats a range whose
 * samples no longer match as not registered (correct result through the general path, slower).  Always unregister before freeing. */
int zkhip_register_bases(const uint64_t *bases, size_t n);
int zkhip_unregister_bases(const uint64_t *bases);

/* ---- NTT: replaces `best_fft(a: &mut [Fr], omega: Fr, log_n: u32)` --------------------------------- */
/* In place, natural order in and out: a[i] <- sum_j a[j] * omega^(i j).  log_n <= 28. */
int zkhip_ntt_fr(uint64_t *a, const uint64_t omega[4], uint32_t log_n);

/* `batch` contiguous polynomials, each transformed in place, one launch set per device.  With several devices (zkhip_init) the batch is
 * cut into contiguous shares, one per device: every transform runs on ONE device (independent polynomials are
 * independent units), every device moves its share over its own PCIe link; results do not depend on the device count. */
int zkhip_ntt_fr_batch(uint64_t *a, const uint64_t omega[4], uint32_t log_n, uint32_t batch);
/* Which batched transforms are spread over the devices: 0 none, 1 (default) the host-buffer forms (zkhip_ntt_fr_batch,
 * zkhip_ifft_scaled_batch, zkhip_coeff_to_extended_batch), 2 also the `_device` batch forms (zkhip_ntt_fr_batch_device,
 * zkhip_ifft_scaled_batch_device, zkhip_coeff_to_extended_device, zkhip_extended_to_coeff_device: a secondary device pulls its
 * polynomials from the primary's HBM over xGMI, transforms them with its own twiddle plan and pushes the results back -- off by
 * default).  $ZKHIP_NTT_FANOUT sets the initial mode. */
int zkhip_set_ntt_fanout(int mode);
int zkhip_ntt_fanout(void);

/* ---- EvaluationDomain pieces ([DEP] halo2_proofs/src/poly/domain.rs), host buffers ----------------- */
/* `EvaluationDomain::ifft`: best_fft with omega_inv, then every element times `divisor`. */
int zkhip_ifft_scaled(uint64_t *a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4]);
/* the same for `batch` contiguous polynomials (`lagrange_to_coeff` of every advice column of a phase in one call) */
int zkhip_ifft_scaled_batch(uint64_t *a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], uint32_t batch);
/* `coeff_to_extended`: a (2^k coeffs) -> out (2^ext_k evaluations on the coset zeta * <ext_omega>):
 * distribute_powers_zeta(into_coset) (a[i] *= {1, zeta, zeta^2}[i % 3]), zero-pad, best_fft. */
int zkhip_coeff_to_extended(const uint64_t *a, uint32_t k, uint64_t *out, uint32_t ext_k, const uint64_t ext_omega[4],
                            const uint64_t zeta[4]);
/* `batch` polynomials: a[b * 2^k ..] -> out[b * 2^ext_k ..] (the per-column loop of the quotient phase in one call) */
int zkhip_coeff_to_extended_batch(const uint64_t *a, uint32_t k, uint64_t *out, uint32_t ext_k, uint32_t batch, const uint64_t ext_omega[4],
                                  const uint64_t zeta[4]);
/* `extended_to_coeff`: ifft on the extended domain, distribute_powers_zeta(out of coset), truncate to
 * out_len elements (= n * quotient_poly_degree).  `a` is consumed (overwritten). */
int zkhip_extended_to_coeff(uint64_t *a, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                            const uint64_t zeta[4], uint64_t *out, size_t out_len);
/* `divide_by_vanishing_poly`: a[i] *= table[i % period] (table = inverted t_evaluations). */
int zkhip_mul_periodic(uint64_t *a, size_t n, const uint64_t *table, uint32_t period);

/* ---- Fr-vector primitives of the prover besides the NTT ------------------------ */
/* `eval_polynomial(poly, point)` [DEP arithmetic.rs]: out = sum poly[i] * point^i. */
int zkhip_fr_eval_polynomial(const uint64_t *poly, size_t n, const uint64_t point[4], uint64_t out[4]);
/* `kate_division(a, b)` [DEP arithmetic.rs]: quotient of a(X) by (X - b), remainder dropped; q has n - 1 elements. */
int zkhip_fr_kate_division(const uint64_t *a, size_t n, const uint64_t b[4], uint64_t *q);
/* `BatchInvert::batch_invert` [DEP ff]: in place, zeros stay zero. */
int zkhip_fr_batch_invert(uint64_t *a, size_t n);
/* grand-product running product [DEP plonk/permutation/prover.rs]: out[0] = 1, out[i] = v[0] * ... * v[i-1] (out may alias v). */
int zkhip_fr_prefix_product(const uint64_t *v, size_t n, uint64_t *out);
int zkhip_fr_eval_polynomial_device(const void *d_poly, size_t n, const uint64_t point[4], void *d_out, void *stream);
int zkhip_fr_kate_division_device(const void *d_a, size_t n, const uint64_t b[4], void *d_q, void *stream);
/* multiopen: `count` device-resident polynomials of n coefficients each (d_polys: host array of device pointers), all evaluated at
 * `point`; d_out receives count results (32 bytes each).  One launch per recursion level for the whole batch. */
int zkhip_fr_eval_polynomial_batch_device(const void *const *d_polys, size_t count, size_t n, const uint64_t point[4], void *d_out, void *stream);
int zkhip_fr_batch_invert_device(void *d_a, size_t n, void *stream);
int zkhip_fr_prefix_product_device(const void *d_v, size_t n, void *d_out, void *stream);

/* ---- row programs: the quotient numerator and every other pointwise pass ---------- */
/* [DEP] halo2_proofs/src/plonk/evaluation.rs evaluates h(X)'s numerator row by row over the extended coset with a small
 * straight-line program per row (`GraphEvaluator`: `Calculation::{Add,Sub,Mul,Square,Double,Negate,Horner,Store}` over
 * `ValueSource::{Constant,Intermediate,Fixed,Advice,Instance,Challenge,Beta,Gamma,Theta,Y,PreviousValue}` with rotations
 * `(row + rot * rot_scale) mod rows`), followed by hand-written permutation / lookup terms of the same shape; reached from
 * create_proof, /root/reference/aggregator/src/wrapper.rs:129.  `zkhip_fr_eval_rows` runs such a program for every row in one
 * fused pass: each thread owns a row, the program is decoded once per wavefront.  The same entry point expresses the
 * multiopen linear combinations, `divide_by_vanishing_poly`, and the numerators / denominators of the grand products.
 *
 * Operands (zkhip_vm_operand):
 *   ZKHIP_SRC_CONST   constants[index]                 (Constant / Challenge / Beta / Gamma / Theta / Y of the reference)
 *   ZKHIP_SRC_REG     register `index` < ZKHIP_VM_REGS (Intermediate; registers start at 0 for every row)
 *   ZKHIP_SRC_COLUMN  columns[index][(row + rotations[rot] * rot_scale) mod rows]      (Fixed / Advice / Instance)
 *   ZKHIP_SRC_PREV    out[row] as it was before the call (PreviousValue); 0 unless `accumulate`
 *   ZKHIP_SRC_ROWPOW  omega^row  (the reference's running `beta_term *= extended_omega`; needs `omega`)
 * Instructions: dst = MOV a | a + b | a - b | a * b | -a | 2a | a^2 | a * b + c.   out[row] = register `result_reg`.
 * rows = 2^log_rows; all field elements are canonical Montgomery Fr words, in and out.  The register file stays in VGPRs, and the
 * library runs the kernel variant sized for the highest register a program names (6 / 8 / 12 / 16: fewer registers = more waves in
 * flight), so a host compiling a `GraphEvaluator` should reuse registers (linear scan over last uses) and number them from 0. */
#define ZKHIP_VM_REGS 16
enum { ZKHIP_SRC_CONST = 0, ZKHIP_SRC_REG = 1, ZKHIP_SRC_COLUMN = 2, ZKHIP_SRC_PREV = 3, ZKHIP_SRC_ROWPOW = 4 };
enum { ZKHIP_OP_MOV = 0, ZKHIP_OP_ADD = 1, ZKHIP_OP_SUB = 2, ZKHIP_OP_MUL = 3, ZKHIP_OP_NEG = 4, ZKHIP_OP_DBL = 5, ZKHIP_OP_SQR = 6,
       ZKHIP_OP_MAD = 7 };
typedef struct zkhip_vm_operand { uint8_t kind; uint8_t rot; uint16_t index; } zkhip_vm_operand;
typedef struct zkhip_vm_insn { uint8_t op; uint8_t dst; uint16_t reserved; zkhip_vm_operand a, b, c; } zkhip_vm_insn;   /* 16 bytes */
typedef struct zkhip_vm_program {
  const zkhip_vm_insn *insns; uint32_t n_insns;
  const uint64_t *constants;  uint32_t n_constants;   /* n_constants x 4 words, host memory */
  const int32_t *rotations;   uint32_t n_rotations;   /* rotation slots, in rows of the base domain */
  int32_t rot_scale;                                  /* 1 on the base domain, 2^(extended_k - k) on the extended one */
  uint32_t result_reg;
  const uint64_t *omega;                              /* 4 words or NULL when ZKHIP_SRC_ROWPOW is not used */
} zkhip_vm_program;
/* columns: n_columns host pointers to 2^log_rows elements each; out: 2^log_rows elements (read first when accumulate != 0) */
int zkhip_fr_eval_rows(const zkhip_vm_program *prog, const uint64_t *const *columns, uint32_t n_columns, uint32_t log_rows,
                       int accumulate, uint64_t *out);
/* same with device-resident columns / output (`d_columns` itself is a host array of device pointers; out may alias a column
 * only if that column is read at rotation 0 exclusively) */
int zkhip_fr_eval_rows_device(const zkhip_vm_program *prog, const void *const *d_columns, uint32_t n_columns, uint32_t log_rows,
                              int accumulate, void *d_out, void *stream);
/* Short programs over many rows (at most 256 instructions, at least 2^18 rows, at most 96 columns) run as straight-line code compiled at
 * run time with hiprtc -- once per (program shape, rows, device), cached; constants stay a table, so one compilation serves every proof of a
 * circuit -- and every other program, or any failure to compile, through the interpreter: same results.  $ZKHIP_VM_JIT = 0 switches it off.
 * zkhip_vm_jit_source returns the generated source (buf may be NULL; *len = bytes needed incl. NUL); zkhip_vm_jit_compile compiles it without
 * launching anything (no device needed), e.g. at keygen, so that the first proof does not pay the seconds of compilation. */
int zkhip_vm_jit_source(const zkhip_vm_program *prog, uint32_t n_columns, uint32_t log_rows, char *buf, size_t cap, size_t *len);
int zkhip_vm_jit_compile(const zkhip_vm_program *prog, uint32_t n_columns, uint32_t log_rows, size_t *code_bytes);
/* out[row] = sum_p weights[p] * progs[p](row): `n_progs` independent row programs over the same columns, run side by side in one launch
 * (one grid row per program) and combined with zkhip_fr_linear_combination_device's kernel.  Programs that read ZKHIP_SRC_ROWPOW must
 * name the same omega.  For programs that are
 * long and run over few rows: the quotient numerator of a circuit with hundreds of columns at 2^13 .. 2^15 rows (the reference's voter and
 * state-transition shapes) is thousands of instructions, and one program there is a handful of wavefronts walking the whole list.  The
 * y-fold of `evaluate_h` is linear in its terms, h = sum_i term_i y^(T-1-i), so a host cuts the term list into consecutive runs, folds each
 * into a program of its own and passes weights[p] = y^(number of terms after run p) (zksnap_circuits_halo2_amd/evaluation.py
 * `evaluate_h_parts`).  `weights`: n_progs x 4 words.  ZKHIP_SRC_PREV reads 0 in every program (there is no previous value). */
int zkhip_fr_eval_rows_sum_device(const zkhip_vm_program *progs, const uint64_t *weights, uint32_t n_progs, const void *const *d_columns, uint32_t n_columns,
                                  uint32_t log_rows, void *d_out, void *stream);
/* out[j] = a[index_a[j]] * b[index_b[j]] (u32 indices, device-resident): the inner loop of `permutation::keygen::Assembly::build_pk`
 * [DEP halo2-axiom plonk/permutation/keygen.rs; keygen_pk at /root/reference/aggregator/src/wrapper.rs:108] -- sigma_i[j] =
 * delta^(column the cell (i, j) maps to) * omega^(its row) -- so that the sigma columns of a proving key are built in HBM.  Indices
 * must be below the table lengths (they are reduced modulo the length rather than trusted: no fault on a bad index). */
int zkhip_fr_gather_mul_device(const void *d_a, size_t a_len, const void *d_index_a, const void *d_b, size_t b_len, const void *d_index_b, size_t n,
                               void *d_out, void *stream);
/* grand product of the permutation / lookup arguments [DEP plonk/permutation/prover.rs, plonk/lookup/prover.rs]:
 * z[0] = 1, z[i+1] = z[i] * num[i] / den[i] for i < n - 1  (zero denominators count as zero, like BatchInvert).
 * num is preserved, den is overwritten (inverted in place), z may alias num. */
int zkhip_fr_grand_product(const uint64_t *num, const uint64_t *den, size_t n, uint64_t *z);
int zkhip_fr_grand_product_device(const void *d_num, void *d_den, size_t n, void *d_z, void *stream);
/* out[i] = sum_j coeffs[j] * d_cols[j][i] for i < n: the y- / v-combinations and L(X) of the multi-open provers over `count` device-resident
 * polynomials ([DEP] poly/kzg/multiopen/shplonk/prover.rs; hundreds of polynomials at the voter / state-transition column counts).  The
 * columns are cut into groups that run side by side, two products under one reduction, so the call does not degrade into `count`
 * dependent multiply-adds per row the way the same sum written as a row program does on few rows.  `d_cols` and `coeffs` (count x 4
 * words, Montgomery) are host arrays; d_out may be one of the columns.  count = 0 gives zeros. */
int zkhip_fr_linear_combination_device(const void *const *d_cols, const uint64_t *coeffs, size_t count, size_t n, void *d_out, void *stream);
/* The KZG multi-open provers for a host that keeps its polynomials as device addresses (a Rust host's handles: rust-shim/prover_patch.rs):
 * `ProverGWC::create_proof` (the reference's gen_snark path, /root/reference/aggregator/src/wrapper.rs:59-60, 127-137) and
 * `ProverSHPLONK::create_proof` (the benches' gen_proof path, /root/reference/aggregator/benches/wrapper_circuit.rs:140)
 * [DEP halo2-axiom poly/kzg/multiopen/{gwc, shplonk}/prover.rs].  A query opens the polynomial of 2^k coefficients at `d_poly` at
 * `point`; `eval` is its value there when `has_eval` is non-zero (the prover has computed and written it to the transcript already:
 * zkhip_fr_eval_polynomial_batch_device), otherwise the library evaluates.  Polynomials are told apart by their addresses.  `bases` is a
 * base array registered with zkhip_register_bases (`ParamsKZG::g`): the commitments are MSMs against its first 2^k points.  The
 * transcript stays with the host: challenges come in, commitments (Jacobian, 12 words each) go out.
 *   GWC      one witness commitment per distinct point, in the order the points first appear among the queries; `capacity` is the room
 *            of `out_points` in points, `*n_out` the number there are (ZKHIP_EINVAL if that is more than `capacity`).
 *   SHPLONK  `begin` (y and v squeezed) returns H = commit(h(X)) and a state; the host writes H, squeezes u; `finish` returns H' and
 *            releases the state whatever it returns (ZKHIP_EINVAL when an evaluation does not belong to its polynomial: the reference's
 *            L(u) = 0 debug assertion).  The polynomials must stay alive and unchanged in between.  `abort` releases a state unused. */
typedef struct zkhip_prover_query {
  uint64_t point[4];
  const void *d_poly;
  uint64_t eval[4];
  uint32_t has_eval;
  uint32_t reserved;
} zkhip_prover_query;   /* 80 bytes */
typedef struct zkhip_shplonk zkhip_shplonk;
int zkhip_multiopen_gwc_device(const uint64_t *bases, uint32_t k, const zkhip_prover_query *queries, size_t n_queries, const uint64_t v[4],
                               uint64_t *out_points, size_t capacity, size_t *n_out);
int zkhip_multiopen_shplonk_begin_device(const uint64_t *bases, uint32_t k, const zkhip_prover_query *queries, size_t n_queries, const uint64_t y[4],
                                         const uint64_t v[4], uint64_t out_h[12], zkhip_shplonk **state);
int zkhip_multiopen_shplonk_finish_device(zkhip_shplonk *state, const uint64_t u[4], uint64_t out_hp[12]);
int zkhip_multiopen_shplonk_abort(zkhip_shplonk *state);
/* The permutation argument's grand products, every set in one call: [DEP] halo2-axiom plonk/permutation/prover.rs `Argument::commit`
 * (the loop over `columns.chunks(chunk_len)`; reached from create_proof, /root/reference/aggregator/src/wrapper.rs:129).  `values[c]` / `sigmas[c]`
 * are the Lagrange values of permutation column c and of its sigma polynomial (2^log_n rows each); set s holds columns
 * [s chunk_len, min((s + 1) chunk_len, n_columns)); `usable_rows` = n - (blinding_factors + 1), `delta` = Fr::DELTA, `omega` the domain's
 * generator.  z is [ceil(n_columns / chunk_len)][2^log_n], dense:
 *     z_0[0] = 1,  z_s[0] = z_(s-1)[usable_rows],  z_s[i + 1] = z_s[i] prod_j (v_j[i] + beta delta^c omega^i + gamma) / (v_j[i] + beta sigma_j[i] + gamma)
 * for i < usable_rows; the rows after usable_rows repeat z_s[usable_rows] (the caller overwrites them with its blinding scalars, as the
 * reference does).  Zero denominators count as zero, like BatchInvert.  A handful of launches whatever the number of sets: the chained
 * products of all sets are one prefix product over the [sets][n] array. */
int zkhip_permutation_products(const uint64_t *const *values, const uint64_t *const *sigmas, uint32_t n_columns, uint32_t chunk_len, uint32_t log_n,
                               size_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4], const uint64_t omega[4],
                               uint64_t *z);
int zkhip_permutation_products_device(const void *const *d_values, const void *const *d_sigmas, uint32_t n_columns, uint32_t chunk_len, uint32_t log_n,
                                      size_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4],
                                      const uint64_t omega[4], void *d_z, void *stream);

/* `permute_expression_pair` of the lookup argument [DEP plonk/lookup/prover.rs]: the first `usable_rows` rows of `input` sorted by
 * canonical value into permuted_input; permuted_table holds, at the first row of every run of equal input values, that value, and at
 * the other rows the table values not consumed this way, ascending, handed out from the last such row backwards (the reference's
 * BTreeMap iteration + Vec::pop).  Rows >= usable_rows of the outputs (the blinding rows) are not written.  ZKHIP_EINVAL when an
 * input value does not occur in the table (the reference's Error::ConstraintSystemFailure).  Outputs must not alias the inputs. */
int zkhip_lookup_permute(const uint64_t *input, const uint64_t *table, size_t usable_rows, uint64_t *permuted_input, uint64_t *permuted_table);
int zkhip_lookup_permute_device(const void *d_input, const void *d_table, size_t usable_rows, void *d_permuted_input, void *d_permuted_table,
                                void *stream);

/* ---- device buffers for a host that does not link HIP itself (handles instead of host slices) ---- */
/* The `_device` entry points below take HIP device pointers so that polynomials stay in HBM from iNTT through commit, extended
 * NTT, quotient and back (PCIe is 8x slower than the NTT kernel: DESIGN.md section 5).  A Rust / C host obtains such pointers
 * here.  Copies run on HIP's default stream (the one the `_device` entry points use when `stream` is NULL) and block:
 * zkhip_download returns when the data is in `dst`, zkhip_upload when `src` may be reused. */
int zkhip_alloc(size_t bytes, void **d_ptr);
int zkhip_free(void *d_ptr);
int zkhip_upload(void *d_dst, const void *src, size_t bytes);
int zkhip_download(void *dst, const void *d_src, size_t bytes);
/* wait for everything this process has queued on the device */
int zkhip_sync(void);
/* wait for the work queued on one stream.  zkhip_upload / zkhip_download are blocking copies on HIP's legacy default stream: a stream
 * created with hipStreamNonBlocking (every library-owned stream, every torch side stream) is NOT ordered against them, so a host that
 * enqueues `_device` calls on such a stream calls this before it reads their results back. */
int zkhip_stream_sync(void *stream);

/* ---- device-resident variants (pointers are HIP device pointers; stream is a hipStream_t; NULL = HIP's default stream 0, ordered
 * with the caller's other default-stream work -- e.g. what torch.cuda.current_stream().cuda_stream is when no stream was set) --- */
/* Used by the pipeline / bench so that polynomials and scalars stay in HBM between calls. */
int zkhip_msm_g1_device(const void *d_scalars, const void *d_bases, size_t n, void *d_out_xyz, void *stream);
int zkhip_msm_g2_device(const void *d_scalars, const void *d_bases, size_t n, void *d_out_xyz, void *stream);
/* prepared (fixed-base) path for device-resident bases: the handle owns the table until released */
int zkhip_prepare_bases_device(const void *d_bases, size_t n, uint64_t *handle);
/* same with an explicit window size (2..20; 0 = automatic) -- experiments and tests of the wide-window path */
int zkhip_prepare_bases_device_c(const void *d_bases, size_t n, int window_bits, uint64_t *handle);
int zkhip_release_bases(uint64_t handle);
/* window size (bits) the handle's table was built for; <= 0 for an unknown handle */
int zkhip_prepared_window_bits(uint64_t handle);
int zkhip_msm_g1_prepared_device(uint64_t handle, size_t offset, const void *d_scalars, size_t n, void *d_out_xyz, void *stream);
/* Device-resident scalars against an array pinned with zkhip_register_bases: `bases` is the HOST pointer the caller would pass to
 * zkhip_msm_g1 (any sub-range of a registered array), the scalars and the result live in HBM.  This is `params.commit(&poly)` for a
 * polynomial that never left the device, against the same tables the host-buffer calls use.  A range that spans several shards (several
 * devices named in zkhip_init, or virtual shards) fans out like the host-buffer call: every secondary device pulls its slice of the scalars
 * from the primary device's HBM (peer copy over xGMI), runs its shard on a stream of its own and returns its 96-byte partial; the fold runs
 * on `stream`, which is the only stream the caller has to wait for (SURVEY.md section 8(e) with the scalars resident in HBM).
 * ZKHIP_EINVAL when the range is not inside a registered array. */
int zkhip_msm_g1_registered_device(const uint64_t *bases, const void *d_scalars, size_t n, void *d_out_xyz, void *stream);
/* `batch` device-resident scalar vectors (vector k at d_scalars + k * scalar_stride elements) against the same registered range, e.g. all
 * advice columns of a circuit; d_out_xyz: batch Jacobian results.  Vectors share launch sets per shard when the shard's table has
 * windows of <= 16 bits. */
int zkhip_msm_g1_registered_batch_device(const uint64_t *bases, const void *d_scalars, size_t n, size_t batch, size_t scalar_stride, void *d_out_xyz,
                                         void *stream);
/* `batch` scalar vectors (vector k at d_scalars + k * scalar_stride elements) against the same prepared bases in one launch
 * set -- e.g. all advice columns of a circuit: small MSMs (k = 13..17) then run at large-MSM throughput.  d_out_xyz: batch
 * Jacobian results, 96 bytes each.  Tables with wide windows (n >= 2^20) do not share a launch set; their vectors run alternately
 * on `stream` and on a high-priority stream the library owns (forked from and joined to `stream` with events), so that one MSM's
 * latency-bound sort and reduction tail run under the next one's accumulation: all results are complete once `stream` has
 * drained, as for any other `_device` call. */
int zkhip_msm_g1_prepared_batch_device(uint64_t handle, size_t offset, const void *d_scalars, size_t n, size_t batch, size_t scalar_stride,
                                       void *d_out_xyz, void *stream);
/* window-size override for experiments (0 = automatic) */
int zkhip_msm_g1_device_c(const void *d_scalars, const void *d_bases, size_t n, void *d_out_xyz, int window_bits, void *stream);
int zkhip_ntt_fr_device(void *d_a, const uint64_t omega[4], uint32_t log_n, void *stream);
int zkhip_ifft_scaled_device(void *d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void *stream);
int zkhip_mul_periodic_device(void *d_a, size_t n, const void *d_table, uint32_t period, void *stream);
/* coset transforms on device-resident polynomials, `batch` of them per launch set (polynomial b at base + b * stride elements;
 * d_out must not overlap d_a).  coeff_to_extended reads 2^k coefficients and writes 2^ext_k evaluations per polynomial;
 * extended_to_coeff reads 2^ext_k evaluations and writes out_len coefficients per polynomial. */
int zkhip_coeff_to_extended_device(const void *d_a, size_t a_stride, uint32_t k, void *d_out, size_t out_stride, uint32_t ext_k, uint32_t batch,
                                   const uint64_t ext_omega[4], const uint64_t zeta[4], void *stream);
int zkhip_extended_to_coeff_device(const void *d_a, size_t a_stride, uint32_t ext_k, const uint64_t ext_omega_inv[4], const uint64_t ext_divisor[4],
                                   const uint64_t zeta[4], void *d_out, size_t out_stride, size_t out_len, uint32_t batch, void *stream);
/* `batch` polynomials of 2^log_n elements, polynomial b at d_a + b * stride elements (stride >= 2^log_n), one launch set:
 * many small transforms (voter / state-transition columns at k = 13..17) run at large-transform throughput */
int zkhip_ntt_fr_batch_device(void *d_a, const uint64_t omega[4], uint32_t log_n, uint32_t batch, size_t stride, void *stream);
int zkhip_ifft_scaled_batch_device(void *d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], uint32_t batch,
                                   size_t stride, void *stream);
/* out = sum of m Jacobian points (multi-GPU: fold of the gathered per-rank partial sums) */
int zkhip_g1_sum_device(const void *d_points_xyz, int m, void *d_out_xyz, void *stream);
int zkhip_g1_sum(const uint64_t *points_xyz, int m, uint64_t out_xyz[12]);
int zkhip_msm_window_bits(size_t n);

/* ---- synthetic inputs + per-phase timing (bench / large tests) ------------------------------------- */
/* d_out[i] = (t0 + i*d) * G as G1Affine, i < n, written to device memory: a seeded stand-in for an SRS whose
 * discrete logs are known, so MSM(a, out) = [sum a_i (t0 + i d)] G can be checked with one scalar multiplication.
 * t0, d: Fr in the usual Montgomery memory format. */
int zkhip_g1_gen_walk_device(const uint64_t t0[4], const uint64_t d[4], size_t n, void *d_out, void *stream);
/* `ParamsKZG::setup` [DEP poly/kzg/commitment.rs] building block: d_out[i] = scalars[i] * G as
 * G1Affine for the BN254 generator G = (1, 2), i < n -- g = [s^i] G and g_lagrange = [L_i(s)] G are two such calls on scalar vectors
 * the Fr primitives above produce on the device.  The first call builds a 32 MiB table of generator multiples. */
int zkhip_g1_fixed_base_mul_device(const void *d_scalars, size_t n, void *d_out, void *stream);
/* `best_fft::<G1>(a, omega, log_n)` [DEP arithmetic.rs: the FftGroup instance for curve points]: in place on 2^log_n Jacobian points
 * (96 B each, the memory of `G1`), natural order in and out, a[i] <- sum_j omega^(i j) a[j]; omega: Fr in Montgomery form.
 * Each butterfly holds one 254-bit scalar multiplication: ~1 s at log_n = 22.  log_n <= 26. */
int zkhip_g1_fft_device(void *d_points_xyz, const uint64_t omega[4], uint32_t log_n, void *stream);
/* `g_to_lagrange(g, k)` [DEP poly/kzg/commitment.rs, used by ParamsKZG::setup / from_parts for an SRS whose trapdoor is not known]:
 * d_g_lagrange[i] = (1/n) sum_j omega_k^(-i j) d_g[j], both arrays 2^k G1Affine points (64 B); may not alias. */
int zkhip_g_to_lagrange_device(const void *d_g, uint32_t k, void *d_g_lagrange, void *stream);
/* host-buffer form with the memory of the reference's function: g_xyz = 2^k Jacobian points (`Vec<G1>`, 12 limbs each), g_lagrange = 2^k affine
 * points (`Vec<G1Affine>`, 8 limbs each); the inverse FFT over the points, the 1/n scaling and the batch normalisation in one call */
int zkhip_g_to_lagrange(const uint64_t *g_xyz, uint32_t k, uint64_t *g_lagrange);
/* `Curve::batch_normalize` [DEP group / halo2curves; create_proof normalises its commitments before they enter the transcript, and
 * keygen stores `to_affine()` of the fixed / permutation commitments in the verifying key]: n Jacobian points (12 limbs each) -> n
 * affine points (8 limbs each, canonical Montgomery limbs; the identity becomes (0, 0)). */
int zkhip_g1_batch_normalize(const uint64_t *points_xyz, size_t n, uint64_t *out_affine);
int zkhip_g1_batch_normalize_device(const void *d_points_xyz, size_t n, void *d_out_affine, void *stream);
/* `G1Affine::read_raw`'s validity check [DEP halo2curves; run on every point by the SerdeFormat::RawBytes readers of SRS and key files:
 * ParamsKZG::read, VerifyingKey::read]: coordinates canonical Montgomery residues and (x, y) = (0, 0) or y^2 = x^3 + 3.  *first_bad =
 * index of the first point that fails, n when all pass (the XYZZ formulas never use b, so an off-curve point would otherwise give
 * garbage commitments without any error). */
int zkhip_g1_check_points(const uint64_t *points, size_t n, uint64_t *first_bad);
int zkhip_g1_check_points_device(const void *d_points, size_t n, uint64_t *first_bad, void *stream);
/* `GroupEncoding::{to_bytes, from_bytes}` of G1Affine [DEP halo2curves derive/curve.rs], one call per point in the reference's
 * `SerdeFormat::Processed` writers / readers (ParamsKZG::{write_custom, read_custom}, VerifyingKey / ProvingKey::{write, read}); the
 * reference itself writes RawBytesUnchecked (/root/reference/aggregator/src/wrapper.rs:971-988).  32 bytes per point: x canonical
 * little-endian plus flags in the last byte.  flag_layout 0 (halo2curves >= 0.3.2): bit 6 = lsb of canonical y, bit 7 = identity;
 * flag_layout 1 (earlier releases): bit 7 = lsb of canonical y, identity = 32 zero bytes.  Decompression is y = (x^3 + 3)^((q+1)/4)
 * per point, strict: *first_bad = index of the first encoding that is not canonical / not on the curve (its output point is (0, 0)),
 * n when all decode. */
int zkhip_g1_compress(const uint64_t *points, size_t n, uint8_t *out32, int flag_layout);
int zkhip_g1_compress_device(const void *d_points, size_t n, void *d_out32, int flag_layout, void *stream);
int zkhip_g1_decompress(const uint8_t *in32, size_t n, uint64_t *points, int flag_layout, uint64_t *first_bad);
int zkhip_g1_decompress_device(const void *d_in32, size_t n, void *d_points, int flag_layout, uint64_t *first_bad, void *stream);
/* Per-phase timing with HIP events on the stream the kernels run on.  enable(1), run one call, then
 * zkhip_profile_read synchronises and returns the number of phases of the last profiled call, writing up to
 * `max` durations (milliseconds) and names (63 chars + NUL each). */
int zkhip_profile_enable(int on);
int zkhip_profile_read(double *ms, char (*names)[64], int max);
/* enable(2): every following call APPENDS its phases (no read-back between calls: a loop of calls runs back to back as it does
 * unprofiled); zkhip_profile_read_calls then synchronises and returns all recorded phases in call order, call_of[i] = index of the
 * call phase i belongs to.  The pool holds 2048 events (about 100 MSM calls); calls beyond it record nothing. */
int zkhip_profile_read_calls(double *ms, char (*names)[64], int *call_of, int max);

/* ---- parity hooks for the field / curve layer (rows a1/a2 of SURVEY.md section 8) ------------------ */
/* field: 0 = Fq, 1 = Fr.  op: 0 mul, 1 add, 2 sub, 3 square (b ignored).  Elementwise on n elements. */
int zkhip_test_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* op: 0 = affine a[i] + affine b[i], 1 = 2 * a[i], 2 = a[i] + (-b[i]); with the quad-cooperative formulas of the reduction tail:
 * 3 = 2 a[i] + 2 b[i], 4 = 4 a[i].  out: n Jacobian points. */
int zkhip_test_g1_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out_xyz, size_t n);

/* G2 (Fq2 + twist curve arithmetic): op 0 = a[i] + b[i], 1 = 2 a[i], 2 = a[i] - b[i]; with the quad-cooperative formulas of the MSM's
 * latency-bound end: 3 = 2 a[i] + b[i], 4 = 4 a[i]; with their lazy forms chained as in the window fold: 5 = 4 a[i] + b[i], 6 = 16 a[i].
 * a, b: n G2Affine points, out: n G2 Jacobian points */
int zkhip_test_g2_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out_xyz, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
