//! Edits to the prover of halo2-axiom [DEP; `halo2_proofs/src/plonk/prover.rs`, `plonk/permutation/prover.rs`, `plonk/lookup/prover.rs`,
//! `plonk/vanishing/prover.rs`, `plonk/evaluation.rs`, `poly/commitment.rs`, `poly/domain.rs`], reached from the reference through
//! `create_proof` (/root/reference/aggregator/src/wrapper.rs:129, `gen_snark`; /root/reference/aggregator/benches/wrapper_circuit.rs:140,
//! `gen_proof`; /root/reference/voter/benches/voter_circuit.rs:80).
//!
//! arithmetic_patch.rs alone already puts every MSM and NTT of an unmodified `create_proof` on the GPU, one call per column.  What that
//! leaves on the table (measured, bench.py `wrapper_replay` / `small_circuit_replays`):
//!   * the small circuits commit hundreds of columns of 2^13..2^15 points one by one -- each call is latency-bound (0.26 ms for 2^13),
//!     256 of them 68 ms where ONE batched call takes 6.0 ms;
//!   * every column crosses PCIe four times (Lagrange values up for the commitment, up again and down for `lagrange_to_coeff`, up again
//!     and 4x down for `coeff_to_extended`, ...): the wrapper's k = 22 call mix is 377 ms through host buffers and 119 ms device-resident.
//! Two modes, both optional, both falling back to the crate's own loops on any non-zero status:
//!   (a) ONE CALL PER PHASE over host buffers -- three defaulted trait / inherent methods and five loop replacements; no new state;
//!   (b) DEVICE-RESIDENT columns -- opt-in (`ZKHIP_DEVICE_RESIDENT=1`): the witness columns go up once, everything from the advice
//!       commitments to the quotient's commitments and the evaluations at x runs on handles (`zkhip_ffi::DevCols`), and only commitments
//!       and evaluations come back for the transcript.  Proof bytes are the same: every value that enters the transcript is the same
//!       field element / group element (tests/cpp/prover_sequence.c checks the three sequences against each other on the GPU).
//! The generic signatures of the crate are kept throughout (`Scheme: CommitmentScheme`, `P: Prover<'params, Scheme>`, `C: CurveAffine`):
//! the new methods are generic with a CPU body, and dispatch to the GPU by `TypeId` inside zkhip_ffi, exactly like `best_multiexp`.
//!
//! NOT COMPILED: no Rust toolchain exists in the image this repository is built in, and the reference pins its dependencies by git branch
//! only (/root/reference/aggregator/Cargo.toml:7-21), so the exact upstream text these edits apply to is not available here.  The loop
//! shapes quoted below are those of halo2-axiom 0.4.x `create_proof`; a maintainer applies them by hand and lets rustc check them.
//! What is checked here: every `zkhip_ffi::` item used below exists (tests/test_rust_shim.py), the `extern "C"` block agrees with
//! include/zkhip.h, the `repr(C)` program structs agree with the header's structs field by field, and the call sequence of both modes
//! runs on the GPU from C99 with every commitment compared (tests/cpp/prover_sequence.c, tests/test_gpu_prover_sequence.py).

use crate::zkhip_ffi::{self, DevCols, VmInsn, VmOperand, VmProgramOwned};

// ======================================================================================================================================
// (a) one call per phase, host buffers
// ======================================================================================================================================

// ---- poly/commitment.rs: `pub trait Params<'params, C: CurveAffine>` gains two DEFAULTED methods (no implementor has to change) -------
//
//     /// Commitments of several Lagrange-basis polynomials (KZG ignores the blind).  Default: one `commit_lagrange` per polynomial.
//     fn commit_lagrange_many(&self, polys: &[&Polynomial<C::ScalarExt, LagrangeCoeff>]) -> Vec<C::CurveExt> {
//         polys.iter().map(|p| self.commit_lagrange(p, Blind::default())).collect()
//     }
//     /// The same in the coefficient basis (`ParamsProver::commit`); lives on `ParamsProver`.
//     fn commit_many(&self, polys: &[&Polynomial<C::ScalarExt, Coeff>]) -> Vec<C::CurveExt> {
//         polys.iter().map(|p| self.commit(p, Blind::default())).collect()
//     }
//
// ---- poly/kzg/commitment.rs: `impl<'params, E: Engine> Params<'params, E::G1Affine> for ParamsKZG<E>` overrides them -------------------
//
//     fn commit_lagrange_many(&self, polys: &[&Polynomial<E::Scalar, LagrangeCoeff>]) -> Vec<E::G1> {
//         let cols: Vec<&[E::Scalar]> = polys.iter().map(|p| &p.values[..]).collect();
//         if let Some(out) = zkhip_ffi::try_msm_g1_many::<E::G1Affine, E::Scalar, E::G1>(&cols, &self.g_lagrange, E::G1::identity()) {
//             return out;
//         }
//         polys.iter().map(|p| self.commit_lagrange(p, Blind::default())).collect()
//     }
//     // `commit_many`: the same with `&self.g`.
//
// ---- poly/domain.rs: `impl<F: WithSmallOrderMulGroup<3>> EvaluationDomain<F>` gains two methods ------------------------------------------
//
//     /// `lagrange_to_coeff` of every polynomial of a phase (one fused call for bn256::Fr; otherwise the per-polynomial loop).
//     pub fn lagrange_to_coeff_many(&self, mut polys: Vec<Polynomial<F, LagrangeCoeff>>) -> Vec<Polynomial<F, Coeff>> {
//         {
//             let mut cols: Vec<&mut [F]> = polys.iter_mut().map(|p| &mut p.values[..]).collect();
//             if zkhip_ffi::try_ifft_scaled_many::<F>(&mut cols, &self.omega_inv, self.k, &self.ifft_divisor) {
//                 return polys.into_iter().map(|p| Polynomial { values: p.values, _marker: PhantomData }).collect();
//             }
//         }
//         polys.into_iter().map(|p| self.lagrange_to_coeff(p)).collect()
//     }
//     /// `coeff_to_extended` of every polynomial the quotient reads.
//     pub fn coeff_to_extended_many(&self, polys: &[&Polynomial<F, Coeff>]) -> Vec<Polynomial<F, ExtendedLagrangeCoeff>> {
//         let cols: Vec<&[F]> = polys.iter().map(|p| &p.values[..]).collect();
//         if let Some(extended) = zkhip_ffi::try_coeff_to_extended_many::<F>(&cols, self.k, self.extended_k, &self.extended_omega, &self.g_coset, F::ZERO) {
//             return extended.into_iter().map(|values| Polynomial { values, _marker: PhantomData }).collect();
//         }
//         polys.iter().map(|p| self.coeff_to_extended(p)).collect()
//     }
//
// ---- plonk/prover.rs `create_proof`: five loop replacements -----------------------------------------------------------------------------
//
//  1. advice commitments of a phase (inside `for current_phase in pk.vk.cs.phases()`):
//         let advice_commitments_projective: Vec<_> = advice_values.iter().zip(blinds.iter())
//             .map(|(poly, blind)| params.commit_lagrange(poly, *blind)).collect();
//     becomes
//         let advice_commitments_projective = params.commit_lagrange_many(&advice_values.iter().collect::<Vec<_>>());
//     (instance commitments, when `P::QUERY_INSTANCE`, the same way.)
//  2. `lookup::Argument::commit_permuted` commits `permuted_input_expression` and `permuted_table_expression` with two
//     `params.commit_lagrange` calls: one `commit_lagrange_many(&[&permuted_input_expression, &permuted_table_expression])`.
//  3. `permutation::Argument::commit` commits the product polynomial of every chunk inside its `for (columns, permutations) in ...chunks`
//     loop: collect the `z` of all chunks first (the chaining `last_z` does not need the commitments), then ONE `commit_lagrange_many`
//     and ONE `domain.lagrange_to_coeff_many`.  `lookup::Permuted::commit_product`: all lookups' `z` in one call each, hoisted out of the
//     `.map(|lookups| ...)` closure of create_proof.
//  4. `advice.advice_polys = advice_values.into_iter().map(|poly| domain.lagrange_to_coeff(poly)).collect()` becomes
//     `domain.lagrange_to_coeff_many(advice_values)`.
//  5. `vanishing::Committed::construct`: the `h_pieces` are committed by `h_pieces.iter().map(|h_piece| params.commit(h_piece, ..))`:
//     `params.commit_many(&h_pieces.iter().collect::<Vec<_>>())`.
//  In `Evaluator::evaluate_h` (plonk/evaluation.rs) the per-column `domain.coeff_to_extended(poly)` maps over `pk.fixed_polys`,
//  `advice_polys`, `instance_polys` become `domain.coeff_to_extended_many(..)`.  (halo2-axiom releases that evaluate the quotient in
//  `extended_len / n` parts call `coeff_to_extended_part` -- an n-point transform of coefficients scaled by powers of zeta * extended_omega^j
//  -- per column and part: those are `best_fft` calls of size n and are batched the same way through `zkhip_ntt_fr_batch`.)

// ======================================================================================================================================
// (b) device-resident columns
// ======================================================================================================================================
//
// State of one proof (plonk/prover.rs, created after witness generation when `std::env::var_os("ZKHIP_DEVICE_RESIDENT").is_some()` and
// the scheme is KZG over bn256 -- `zkhip_ffi::is::<Scheme::Curve, G1Affine>()`):

/// Column order of the quotient's row program = `evaluation.py quotient_columns` of this repository: fixed | advice | instance |
/// l_0, l_last, l_active_row | permutation sigma | permutation products (one per set) | per lookup: product, permuted input, permuted table.
pub(crate) struct DeviceProof {
    /// Lagrange-basis values, later (in place) coefficients, of every column: `n` rows each
    pub base: DevCols,
    /// the same columns on the extended coset: `extended_len` rows each.  The proving key's columns (fixed, l_*, sigma) are uploaded and
    /// transformed ONCE per `ProvingKey` and kept (`pk.ev` gains an `Option<Arc<DeviceKey>>` filled on first use); only the
    /// witness-dependent columns are transformed per proof
    pub ext: DevCols,
    /// scratch: one column of `n` rows (grand-product denominators), two of `extended_len` rows (h evaluations, h coefficients)
    pub den: DevCols,
    pub h: DevCols,
    pub first_advice: usize,
    pub first_perm_product: usize,
    pub first_lookup: usize,
}

// The order of `create_proof` is unchanged; each step that used host `Polynomial`s uses the handle instead:
//
//   advice (per phase)     base.upload(col, 0, &advice_values[i]) for the phase's columns; commitments = base.commit_many(first, count, n,
//                          &params.g_lagrange, identity) -> batch_normalize -> transcript (unchanged from here)
//   lookups, theta         compressed input / table expressions = DevCols::eval_rows(&lower_expression(..)) over base columns (a lookup of plain
//                          columns needs no program); base.lookup_permute(..) writes the permuted pair; blinding rows:
//                          base.upload(col, usable_rows, &random_rows); commit_many of the pair
//   permutation, beta/gamma  base.permutation_products(first_perm_product, &value_addresses, &sigma_addresses, chunk_len, k, usable_rows, &beta, &gamma,
//                          &F::DELTA, &omega): every chunk's z, chained through z[usable_rows], in one call (a handful of launches whatever the
//                          number of chunks: hundreds at the voter / state-transition column counts); blinding rows uploaded; commit_many over
//                          all chunks.  (Before this entry point existed: per chunk a numerator and a denominator program,
//                          base.grand_product, a 32-byte download and a scaling program.)
//   lookup products        (a' + beta)(s' + gamma) denominators, (compressed input + beta)(compressed table + gamma) numerators, grand_product
//   vanishing random poly  generated on the host as upstream, committed through best_multiexp (one column; stays on the host path)
//   y                      base.ifft_scaled_many(first_advice, all witness-dependent columns, ..): lagrange_to_coeff in place, one call
//   quotient               base.coeff_to_extended_many(.., &ext, ..) for the witness-dependent columns; DevCols::eval_rows(&quotient_program, all
//                          columns of ext, extended_k, &h, 0); h.quotient_to_coeff(..) (divide_by_vanishing_poly + extended_to_coeff);
//                          the three pieces = h.commit_many(1, ..) with stride n over column 1 (coefficients), against &params.g
//                          For a circuit with hundreds of columns at k = 13 .. 15 (the voter and the state transition) the quotient program is
//                          thousands of instructions over a few thousand rows: cut the evaluator's calculations into ~16 runs at `Store`
//                          boundaries (the y-fold is linear: h = sum_i term_i y^(T-1-i)), lower each run, and call
//                          DevCols::eval_rows_sum(&programs, &weights, ..) with weights[p] = y^(terms after run p) -- 4.8 -> 1.2 ms at 947 columns
//   x                      DevCols::eval_polys(addresses of every opened polynomial, n, &x, F::ZERO) -> transcript
//   multiopen              the opened polynomials stay on the device: instead of `ProverQuery { point, poly: &Polynomial, blind }` the prover
//                          collects `zkhip_ffi::dev_query(&point, base.at(column, 0), &eval)` in the order upstream builds its queries
//                          (instances, advice, permutation, lookups, fixed, vanishing) and calls `open_on_device` below in place of
//                          `P::create_proof(rng, transcript, queries)`.  Which argument runs is the scheme's: GWC on the gen_snark path
//                          (/root/reference/aggregator/src/wrapper.rs:127-129), SHPLONK on the benches' (halo2-base `gen_proof`).  When the
//                          device path declines (None), the polynomials are downloaded once (base.download) and the upstream prover runs.

/// What poly/kzg/multiopen/{gwc, shplonk}/prover.rs do with the transcript, around the device-side provers: the SAME squeezes and writes
/// in the SAME order as upstream (GWC: v, then one witness per point; SHPLONK: y, v, H, u, H'), so the proof bytes do not change.
/// `T` is the transcript (`TranscriptWrite<C, E>`), `squeeze` / `write` are its `squeeze_challenge_scalar` / `write_point` closures --
/// passed in because their trait bounds are the patched file's, not this module's.
pub(crate) enum OpenWith { Gwc, Shplonk }
pub(crate) fn open_on_device<C: 'static, F: 'static, P: 'static + Clone>(which: OpenWith, g: &[C], k: u32, queries: &[zkhip_ffi::ProverQueryC], identity: P,
                                                                         squeeze: &mut dyn FnMut() -> F, write: &mut dyn FnMut(&P) -> bool) -> Option<()> {
    match which {
        OpenWith::Gwc => {
            let v = squeeze();
            for w in zkhip_ffi::multiopen_gwc(g, k, queries, &v, identity)? { if !write(&w) { return None; } }
        }
        OpenWith::Shplonk => {
            let (y, v) = (squeeze(), squeeze());
            let (session, h) = zkhip_ffi::ShplonkSession::begin(g, k, queries, &y, &v, identity.clone())?;
            if !write(&h) { return None; }                      // (the session drops and releases its state)
            let u = squeeze();
            let hp = session.finish(&u, identity)?;
            if !write(&hp) { return None; }
        }
    }
    Some(())
}
// NOTE on `None` after the first squeeze: the transcript has advanced, so the caller must not re-run the upstream prover on the same
// transcript; treat it as the proof failing (Error::Opening upstream), exactly as a failed MSM inside the upstream prover would.

/// `GraphEvaluator` -> row program (plonk/evaluation.rs).  `calcs[i]` writes intermediate `i`; sources are mapped by the closures the
/// caller passes, because the column indices depend on the `DeviceProof` layout above:
///   Calculation::{Add(a,b), Sub(a,b), Mul(a,b), Square(a), Double(a), Negate(a), Horner(start, parts, factor), Store(a)}
///   ValueSource::{Constant(i), Intermediate(i), Fixed(c, r), Advice(c, r), Instance(c, r), Challenge(i), Beta(), Gamma(), Theta(), Y(), PreviousValue()}
/// `rotations` are the evaluator's rotation table (in rows of the base domain; the program's `rot_scale` = 2^(extended_k - k) turns them into
/// rows of the extended coset, or 1 when the quotient is evaluated part by part).  Intermediates get registers by linear scan over last uses
/// (the library runs the kernel variant sized for the highest register named, so fewer registers = more wavefronts in flight); an expression
/// with more than `VM_REGS` simultaneously live intermediates is split by the caller at a `Store` (accumulate through SRC_PREV).
pub(crate) enum Src { Constant(usize), Intermediate(usize), Column { column: usize, rotation_slot: usize }, Previous }
pub(crate) enum Calc { Add(Src, Src), Sub(Src, Src), Mul(Src, Src), Square(Src), Double(Src), Negate(Src), Horner(Src, Vec<Src>, Src), Store(Src) }

pub(crate) fn lower_graph(calcs: &[Calc], constants: Vec<[u64; 4]>, rotations: Vec<i32>, rot_scale: i32, result: usize) -> Option<VmProgramOwned> {
    // every Calc becomes 1 instruction, a Horner `parts.len()` multiply-adds; `slots[i]` = index of the instruction that defines intermediate i
    let operand = |s: &Src, reg_of: &[Option<u8>]| -> Option<VmOperand> {
        Some(match s {
            Src::Constant(i) => VmOperand { kind: zkhip_ffi::SRC_CONST, rot: 0, index: *i as u16 },
            Src::Intermediate(i) => VmOperand { kind: zkhip_ffi::SRC_REG, rot: 0, index: reg_of[*i]? as u16 },
            Src::Column { column, rotation_slot } => VmOperand { kind: zkhip_ffi::SRC_COLUMN, rot: *rotation_slot as u8, index: *column as u16 },
            Src::Previous => VmOperand { kind: zkhip_ffi::SRC_PREV, rot: 0, index: 0 },
        })
    };
    let sources = |c: &Calc| -> Vec<&Src> {
        match c {
            Calc::Add(a, b) | Calc::Sub(a, b) | Calc::Mul(a, b) => vec![a, b],
            Calc::Square(a) | Calc::Double(a) | Calc::Negate(a) | Calc::Store(a) => vec![a],
            Calc::Horner(s, parts, f) => { let mut v = vec![s, f]; v.extend(parts.iter()); v }
        }
    };
    // last use of every intermediate
    let mut last_use = vec![0usize; calcs.len()];
    for (pos, c) in calcs.iter().enumerate() {
        for s in sources(c) { if let Src::Intermediate(i) = s { last_use[*i] = pos; } }
    }
    last_use[result] = calcs.len();
    let mut free: Vec<u8> = (0..zkhip_ffi::VM_REGS as u8).rev().collect();
    let mut reg_of: Vec<Option<u8>> = vec![None; calcs.len()];
    let mut insns: Vec<VmInsn> = Vec::new();
    for (pos, c) in calcs.iter().enumerate() {
        // operands are resolved first.  A one-instruction calculation reads its sources and writes its destination in the same
        // instruction, so registers whose last use is here are released BEFORE the destination is chosen (it may reuse one of them);
        // a Horner is several instructions that keep reading its parts: its destination is chosen first, its sources released after.
        let ops: Vec<VmOperand> = sources(c).into_iter().map(|s| operand(s, &reg_of)).collect::<Option<Vec<_>>>()?;
        let multi = matches!(c, Calc::Horner(..));
        let mut dst = if multi { free.pop()? } else { 0 };
        for s in sources(c) {
            if let Src::Intermediate(i) = s { if last_use[*i] == pos { if let Some(r) = reg_of[*i].take() { free.push(r); } } }
        }
        if !multi { dst = free.pop()?; }                                  // None: more than VM_REGS live intermediates -> the caller splits
        reg_of[pos] = Some(dst);
        let z = VmOperand::default();
        match c {
            Calc::Add(..) => insns.push(VmInsn { op: zkhip_ffi::OP_ADD, dst, reserved: 0, a: ops[0], b: ops[1], c: z }),
            Calc::Sub(..) => insns.push(VmInsn { op: zkhip_ffi::OP_SUB, dst, reserved: 0, a: ops[0], b: ops[1], c: z }),
            Calc::Mul(..) => insns.push(VmInsn { op: zkhip_ffi::OP_MUL, dst, reserved: 0, a: ops[0], b: ops[1], c: z }),
            Calc::Square(..) => insns.push(VmInsn { op: zkhip_ffi::OP_SQR, dst, reserved: 0, a: ops[0], b: z, c: z }),
            Calc::Double(..) => insns.push(VmInsn { op: zkhip_ffi::OP_DBL, dst, reserved: 0, a: ops[0], b: z, c: z }),
            Calc::Negate(..) => insns.push(VmInsn { op: zkhip_ffi::OP_NEG, dst, reserved: 0, a: ops[0], b: z, c: z }),
            Calc::Store(..) => insns.push(VmInsn { op: zkhip_ffi::OP_MOV, dst, reserved: 0, a: ops[0], b: z, c: z }),
            Calc::Horner(_, parts, _) => {
                // value = start; for part in parts { value = value * factor + part }   (ops = [start, factor, parts...])
                let me = VmOperand { kind: zkhip_ffi::SRC_REG, rot: 0, index: dst as u16 };
                insns.push(VmInsn { op: zkhip_ffi::OP_MOV, dst, reserved: 0, a: ops[0], b: z, c: z });
                for p in 0..parts.len() {
                    insns.push(VmInsn { op: zkhip_ffi::OP_MAD, dst, reserved: 0, a: me, b: ops[1], c: ops[2 + p] });
                }
            }
        }
    }
    Some(VmProgramOwned { insns, constants, rotations, rot_scale, result_reg: reg_of[result]? as u32, omega: None })
}

// plonk/evaluation.rs `Evaluator::evaluate_h`, device-resident branch (first statement of the function):
//
//     if let Some(dp) = device_proof {                 // Option<&DeviceProof> threaded from create_proof
//         // custom gates: the evaluator's calculations, sources mapped to dp's column order; y-folding is already part of the graph
//         // (`Calculation::Horner(PreviousValue, gate polynomials, Y)`); the hand-written permutation and lookup terms of this function are
//         // appended as calculations of the same graph (they are sums of products of columns, l_0 / l_last / l_active_row, beta, gamma,
//         // theta and `beta * zeta * delta^j * extended_omega^row` = SRC_ROWPOW times a constant) -- evaluation.py evaluate_h_program of this
//         // repository spells the term list out in upstream's order.
//         let prog = lower_graph(&calcs, constants, rotations, 1 << (domain.extended_k() - domain.k()), result)?;
//         let cols: Vec<*const c_void> = (0..dp.ext.count).map(|c| dp.ext.at(c, 0) as *const c_void).collect();
//         DevCols::eval_rows(&prog, &cols, domain.extended_k(), &dp.h, 0);
//         // the caller (vanishing::Committed::construct) then calls dp.h.quotient_to_coeff(..) instead of
//         // domain.divide_by_vanishing_poly + domain.extended_to_coeff
//     }
