//! `halo2_proofs/src/zkhip_ffi.rs` -- the extern "C" surface of libzkhip.so that the patched halo2-axiom crate binds, plus the
//! type dispatch that keeps the crate's generic signatures (`best_multiexp<C: CurveAffine>`, `best_fft<Scalar, G: FftGroup<Scalar>>`,
//! `ParamsKZG<E: Engine>`) intact: the GPU path is taken when -- and only when -- the instantiation is BN254 (`bn256::G1Affine` /
//! `bn256::Fr`), every other instantiation and every non-zero status falls through to the crate's own CPU body (SURVEY.md section 8(b)
//! "Errors": the reference functions are infallible; the C ABI never aborts).
//!
//! Reached from the reference through `create_proof` (/root/reference/aggregator/src/wrapper.rs:129), `keygen_vk` / `keygen_pk`
//! (wrapper.rs:107-108), `gen_srs` (/root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69) and `gen_proof`
//! (wrapper_circuit.rs:140).  The declarations below are checked against include/zkhip.h by tests/test_rust_shim.py (name, arity,
//! pointer / integer width of every parameter), and the call sequence this module issues is replayed in C99 by
//! tests/cpp/shim_sequence.c on the GPU.
//!
//! Add `pub(crate) mod zkhip_ffi;` to halo2_proofs/src/lib.rs.  No Rust toolchain exists in the image this repository is built in:
//! these files are the binding a maintainer adds, kept compile-ready by review, not by rustc.

use std::any::TypeId;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::atomic::{AtomicU8, Ordering};
use std::sync::Once;

use group::{prime::PrimeCurveAffine, Curve, Group};
use halo2curves::bn256::{Fr, G1Affine, G1};

#[allow(dead_code)]
extern "C" {
    fn zkhip_init(devices: *const c_int, ndev: c_int) -> c_int;
    fn zkhip_last_error() -> *const c_char;
    fn zkhip_msm_g1(scalars: *const u64, bases: *const u64, n: usize, out_xyz: *mut u64) -> c_int;
    fn zkhip_msm_g1_batch(scalars: *const u64, bases: *const u64, n: usize, batch: usize, out_xyz: *mut u64) -> c_int;
    fn zkhip_ntt_fr(a: *mut u64, omega: *const u64, log_n: u32) -> c_int;
    fn zkhip_ntt_fr_batch(a: *mut u64, omega: *const u64, log_n: u32, batch: u32) -> c_int;
    fn zkhip_ifft_scaled(a: *mut u64, omega_inv: *const u64, log_n: u32, divisor: *const u64) -> c_int;
    fn zkhip_coeff_to_extended(a: *const u64, k: u32, out: *mut u64, ext_k: u32, ext_omega: *const u64, zeta: *const u64) -> c_int;
    fn zkhip_extended_to_coeff(a: *mut u64, ext_k: u32, ext_omega_inv: *const u64, ext_divisor: *const u64, zeta: *const u64,
                               out: *mut u64, out_len: usize) -> c_int;
    fn zkhip_mul_periodic(a: *mut u64, n: usize, table: *const u64, period: u32) -> c_int;
    fn zkhip_g_to_lagrange(g_xyz: *const u64, k: u32, g_lagrange: *mut u64) -> c_int;
    fn zkhip_register_bases(bases: *const u64, n: usize) -> c_int;
    fn zkhip_unregister_bases(bases: *const u64) -> c_int;
    // ---- prover_patch.rs, mode (a): one call per phase over host buffers --------------------------------------------------------------
    fn zkhip_ifft_scaled_batch(a: *mut u64, omega_inv: *const u64, log_n: u32, divisor: *const u64, batch: u32) -> c_int;
    fn zkhip_coeff_to_extended_batch(a: *const u64, k: u32, out: *mut u64, ext_k: u32, batch: u32, ext_omega: *const u64, zeta: *const u64) -> c_int;
    // ---- prover_patch.rs, mode (b): device-resident columns (handles instead of host slices) ------------------------------------------
    fn zkhip_alloc(bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    fn zkhip_free(d_ptr: *mut c_void) -> c_int;
    fn zkhip_upload(d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    fn zkhip_download(dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    fn zkhip_msm_g1_registered_batch_device(bases: *const u64, d_scalars: *const c_void, n: usize, batch: usize, scalar_stride: usize,
                                            d_out_xyz: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_ifft_scaled_batch_device(d_a: *mut c_void, omega_inv: *const u64, log_n: u32, divisor: *const u64, batch: u32, stride: usize,
                                      stream: *mut c_void) -> c_int;
    fn zkhip_coeff_to_extended_device(d_a: *const c_void, a_stride: usize, k: u32, d_out: *mut c_void, out_stride: usize, ext_k: u32, batch: u32,
                                      ext_omega: *const u64, zeta: *const u64, stream: *mut c_void) -> c_int;
    fn zkhip_extended_to_coeff_device(d_a: *const c_void, a_stride: usize, ext_k: u32, ext_omega_inv: *const u64, ext_divisor: *const u64,
                                      zeta: *const u64, d_out: *mut c_void, out_stride: usize, out_len: usize, batch: u32, stream: *mut c_void) -> c_int;
    fn zkhip_mul_periodic_device(d_a: *mut c_void, n: usize, d_table: *const c_void, period: u32, stream: *mut c_void) -> c_int;
    fn zkhip_fr_eval_rows_device(prog: *const VmProgram, d_columns: *const *const c_void, n_columns: u32, log_rows: u32, accumulate: c_int,
                                 d_out: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_fr_grand_product_device(d_num: *const c_void, d_den: *mut c_void, n: usize, d_z: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_fr_eval_rows_sum_device(progs: *const VmProgram, weights: *const u64, n_progs: u32, d_columns: *const *const c_void, n_columns: u32, log_rows: u32,
                                     d_out: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_fr_linear_combination_device(d_cols: *const *const c_void, coeffs: *const u64, count: usize, n: usize, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_multiopen_gwc_device(bases: *const u64, k: u32, queries: *const ProverQueryC, n_queries: usize, v: *const u64, out_points: *mut u64,
                                  capacity: usize, n_out: *mut usize) -> c_int;
    fn zkhip_multiopen_shplonk_begin_device(bases: *const u64, k: u32, queries: *const ProverQueryC, n_queries: usize, y: *const u64, v: *const u64,
                                            out_h: *mut u64, state: *mut *mut ShplonkState) -> c_int;
    fn zkhip_multiopen_shplonk_finish_device(state: *mut ShplonkState, u: *const u64, out_hp: *mut u64) -> c_int;
    fn zkhip_multiopen_shplonk_abort(state: *mut ShplonkState) -> c_int;
    fn zkhip_permutation_products_device(d_values: *const *const c_void, d_sigmas: *const *const c_void, n_columns: u32, chunk_len: u32, log_n: u32,
                                         usable_rows: usize, beta: *const u64, gamma: *const u64, delta: *const u64, omega: *const u64, d_z: *mut c_void,
                                         stream: *mut c_void) -> c_int;
    fn zkhip_lookup_permute_device(d_input: *const c_void, d_table: *const c_void, usable_rows: usize, d_permuted_input: *mut c_void,
                                   d_permuted_table: *mut c_void, stream: *mut c_void) -> c_int;
    fn zkhip_fr_eval_polynomial_batch_device(d_polys: *const *const c_void, count: usize, n: usize, point: *const u64, d_out: *mut c_void,
                                             stream: *mut c_void) -> c_int;
}

/// `zkhip_vm_operand` / `zkhip_vm_insn` / `zkhip_vm_program` of include/zkhip.h (field order and widths checked by tests/test_rust_shim.py).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub(crate) struct VmOperand { pub kind: u8, pub rot: u8, pub index: u16 }
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub(crate) struct VmInsn { pub op: u8, pub dst: u8, pub reserved: u16, pub a: VmOperand, pub b: VmOperand, pub c: VmOperand }
#[repr(C)]
pub(crate) struct VmProgram {
    pub insns: *const VmInsn, pub n_insns: u32,
    pub constants: *const u64, pub n_constants: u32,
    pub rotations: *const i32, pub n_rotations: u32,
    pub rot_scale: i32,
    pub result_reg: u32,
    pub omega: *const u64,
}
/// `zkhip_prover_query`: one opening -- the polynomial of 2^k coefficients at device address `d_poly`, opened at `point`; `eval` is its value
/// there when `has_eval != 0` (the prover has written it to the transcript already), otherwise the library evaluates
#[repr(C)]
#[derive(Clone, Copy)]
pub(crate) struct ProverQueryC { pub point: [u64; 4], pub d_poly: *const c_void, pub eval: [u64; 4], pub has_eval: u32, pub reserved: u32 }
/// `zkhip_shplonk`: the state between the two steps of the SHPLONK prover (opaque)
#[repr(C)]
pub(crate) struct ShplonkState { _opaque: [u8; 0] }
pub(crate) const VM_REGS: usize = 16;
pub(crate) const SRC_CONST: u8 = 0;
pub(crate) const SRC_REG: u8 = 1;
pub(crate) const SRC_COLUMN: u8 = 2;
pub(crate) const SRC_PREV: u8 = 3;
pub(crate) const SRC_ROWPOW: u8 = 4;
pub(crate) const OP_MOV: u8 = 0;
pub(crate) const OP_ADD: u8 = 1;
pub(crate) const OP_SUB: u8 = 2;
pub(crate) const OP_MUL: u8 = 3;
pub(crate) const OP_NEG: u8 = 4;
pub(crate) const OP_DBL: u8 = 5;
pub(crate) const OP_SQR: u8 = 6;
pub(crate) const OP_MAD: u8 = 7;

/// Below these sizes a call is answered by the crate's CPU body: a GPU call costs ~0.1 ms of launch chain and PCIe latency whatever its size, and the
/// verifier / accumulation side of the reference (`verify_proof`, `AccumulatorStrategy`, /root/reference/aggregator/src/wrapper.rs:145) issues
/// multi-exponentiations of a handful of points.  (A 2^10-point MSM is ~0.1 ms on one CPU core; the smallest circuit of the reference commits 2^13.)
const MIN_GPU_MSM: usize = 1 << 10;
const MIN_GPU_NTT_LOG: u32 = 11;

/// 0 = not probed, 1 = GPU path usable, 2 = disabled (no device, layout self-test failed, or ZKHIP_DISABLE=1)
static STATE: AtomicU8 = AtomicU8::new(0);
static PROBE: Once = Once::new();
static WARN: Once = Once::new();

fn last_error() -> String {
    // SAFETY: zkhip_last_error returns a NUL-terminated string owned by the library (thread-local buffer), never NULL
    unsafe { std::ffi::CStr::from_ptr(zkhip_last_error()) }.to_string_lossy().into_owned()
}

fn warn_once(what: &str, rc: c_int) {
    WARN.call_once(|| eprintln!("zkhip: {what} failed (status {rc}: {}); using the CPU path", last_error()));
}

/// One-time probe: initialise the library and check, through one MSM of one point, that `Fr`, `G1Affine` and `G1` have the memory
/// layout the C ABI assumes (4 / 8 / 12 little-endian u64 limbs in Montgomery form, fields in declaration order): [1] G must come back
/// as G.  A layout surprise (a halo2curves release that reorders or re-encodes a field) disables the GPU path instead of corrupting proofs.
fn usable() -> bool {
    PROBE.call_once(|| {
        let mut ok = std::env::var_os("ZKHIP_DISABLE").is_none()
            && std::mem::size_of::<Fr>() == 32
            && std::mem::size_of::<G1Affine>() == 64
            && std::mem::size_of::<G1>() == 96
            && std::mem::align_of::<G1>() <= 8;
        if ok {
            // SAFETY: NULL / 0 selects the default device list; the call has no other preconditions
            ok = unsafe { zkhip_init(std::ptr::null(), 0) } == 0;
        }
        if ok {
            let one = [<Fr as ff::Field>::ONE];
            let g = [G1Affine::generator()];
            let mut out = G1::identity();
            // SAFETY: one scalar, one base, one result -- the sizes were checked above
            let rc = unsafe { zkhip_msm_g1(one.as_ptr() as *const u64, g.as_ptr() as *const u64, 1, &mut out as *mut G1 as *mut u64) };
            ok = rc == 0 && out.to_affine() == g[0];
        }
        STATE.store(if ok { 1 } else { 2 }, Ordering::Release);
    });
    STATE.load(Ordering::Acquire) == 1
}

#[inline]
fn is<T: 'static, U: 'static>() -> bool {
    TypeId::of::<T>() == TypeId::of::<U>()
}

/// `best_multiexp::<C>`: Some(result) when C = bn256::G1Affine and the GPU call succeeded, None otherwise (the caller runs its CPU body).
/// `C`, `S` (= C::Scalar) and `P` (= C::Curve) are type parameters so that this module does not need the crate's trait imports; all three
/// carry `'static` through `CurveAffine` / `ff::Field` / `group::Group` [DEP halo2curves, ff, group].
pub(crate) fn try_msm_g1<C: 'static, S: 'static, P: 'static>(coeffs: &[S], bases: &[C], identity: P) -> Option<P> {
    if !(is::<C, G1Affine>() && is::<S, Fr>() && is::<P, G1>()) || coeffs.len() != bases.len() || coeffs.len() < MIN_GPU_MSM || !usable() {
        return None;
    }
    let mut out = identity;
    // SAFETY: the TypeId checks make &[S] = &[Fr] (32-byte elements), &[C] = &[G1Affine] (64) and P = G1 (96 bytes); the library borrows
    // the slices for the duration of the call and writes 12 limbs through `out`
    let rc = unsafe { zkhip_msm_g1(coeffs.as_ptr() as *const u64, bases.as_ptr() as *const u64, coeffs.len(), &mut out as *mut P as *mut u64) };
    if rc != 0 {
        warn_once("zkhip_msm_g1", rc);
        return None;
    }
    Some(out)
}

/// `best_fft::<Scalar, G>`: true when G = Scalar = bn256::Fr and the transform was done in place on the GPU.
pub(crate) fn try_ntt_fr<S: 'static, G: 'static>(a: &mut [G], omega: &S, log_n: u32) -> bool {
    if !(is::<G, Fr>() && is::<S, Fr>()) || log_n < MIN_GPU_NTT_LOG || log_n > 28 || a.len() != 1usize << log_n || !usable() {
        return false;
    }
    // SAFETY: &mut [G] = &mut [Fr], 2^log_n elements of 4 limbs, transformed in place; omega: 4 limbs, read only
    let rc = unsafe { zkhip_ntt_fr(a.as_mut_ptr() as *mut u64, omega as *const S as *const u64, log_n) };
    if rc != 0 {
        // `a` is written only by the final device-to-host copy of a successful transform (capi.hip: host_transform), so the CPU body below
        // still sees the caller's input
        warn_once("zkhip_ntt_fr", rc);
        return false;
    }
    true
}

/// `Pinned::new` (commitment_patch.rs; reached from `ParamsKZG::{setup, read, read_custom, from_parts, clone, downsize}`): pin `g` / `g_lagrange` on the device(s) and build the fixed-base
/// tables, so that `commit` / `commit_lagrange` (which pass `&self.g[..poly.len()]`) upload 32 n bytes of scalars and nothing else.
/// No-op for every curve but bn256::G1Affine.  The memory must stay alive and unchanged until `unpin` (`Drop for Pinned` in commitment_patch.rs).
pub(crate) fn pin<C: 'static>(bases: &[C]) {
    if !is::<C, G1Affine>() || bases.is_empty() || !usable() {
        return;
    }
    // SAFETY: &[C] = &[G1Affine]; the owner (ParamsKZG) keeps the Vec alive and never mutates it while it is registered
    let rc = unsafe { zkhip_register_bases(bases.as_ptr() as *const u64, bases.len()) };
    if rc != 0 {
        warn_once("zkhip_register_bases", rc);   // unregistered bases still work: zkhip_msm_g1 uploads them per call
    }
}

/// `g_to_lagrange::<C>(g_projective, k)` [DEP poly/kzg/commitment.rs; `ParamsKZG::{setup, from_parts, downsize}`, reached from `gen_srs`,
/// /root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69]: the inverse FFT over the 2^k points, the 1/n scaling and the batch normalisation
/// in one call (1.2 s at k = 22, where the CPU body runs 2^22 scalar multiplications).  Some(g_lagrange) for C = bn256::G1Affine, None otherwise.
pub(crate) fn try_g_to_lagrange<C: 'static + Clone, P: 'static>(g_projective: &[P], k: u32, identity: C) -> Option<Vec<C>> {
    if !(is::<C, G1Affine>() && is::<P, G1>()) || k > 26 || k < MIN_GPU_NTT_LOG || g_projective.len() != 1usize << k || !usable() {
        return None;
    }
    let mut out = vec![identity; 1usize << k];
    // SAFETY: &[P] = &[G1] (96-byte Jacobian points, read only), out = Vec<G1Affine> of 2^k elements (64 bytes each), written in full on success
    let rc = unsafe { zkhip_g_to_lagrange(g_projective.as_ptr() as *const u64, k, out.as_mut_ptr() as *mut u64) };
    if rc != 0 {
        warn_once("zkhip_g_to_lagrange", rc);
        return None;
    }
    Some(out)
}

/// Undo `pin`; must run before the memory is freed or rewritten (`Drop for Pinned`, `Pinned::truncate` in commitment_patch.rs).
pub(crate) fn unpin<C: 'static>(bases: &[C]) {
    if !is::<C, G1Affine>() || bases.is_empty() || STATE.load(Ordering::Acquire) != 1 {
        return;
    }
    // SAFETY: by address only; an address that was never registered is reported as ZKHIP_EINVAL and ignored
    let _ = unsafe { zkhip_unregister_bases(bases.as_ptr() as *const u64) };
}

/// `EvaluationDomain::ifft` for F = bn256::Fr: inverse transform and the 1/n scale in one call.
pub(crate) fn try_ifft_scaled<F: 'static>(a: &mut [F], omega_inv: &F, log_n: u32, divisor: &F) -> bool {
    if !is::<F, Fr>() || log_n < MIN_GPU_NTT_LOG || log_n > 28 || a.len() != 1usize << log_n || !usable() {
        return false;
    }
    // SAFETY: as try_ntt_fr
    let rc = unsafe { zkhip_ifft_scaled(a.as_mut_ptr() as *mut u64, omega_inv as *const F as *const u64, log_n, divisor as *const F as *const u64) };
    if rc != 0 {
        warn_once("zkhip_ifft_scaled", rc);
    }
    rc == 0
}

/// `EvaluationDomain::coeff_to_extended` for F = bn256::Fr: zeta-powers, zero padding and the extended transform fused.
/// `a`: 2^k coefficients; `out`: 2^ext_k elements (the caller allocates, as the crate does with `resize`).
pub(crate) fn try_coeff_to_extended<F: 'static>(a: &[F], k: u32, out: &mut [F], ext_k: u32, ext_omega: &F, zeta: &F) -> bool {
    if !is::<F, Fr>() || ext_k > 28 || k > ext_k || a.len() != 1usize << k || out.len() != 1usize << ext_k || !usable() {
        return false;
    }
    // SAFETY: sizes checked above; `a` read only, `out` written only, they do not overlap (distinct borrows)
    let rc = unsafe {
        zkhip_coeff_to_extended(a.as_ptr() as *const u64, k, out.as_mut_ptr() as *mut u64, ext_k, ext_omega as *const F as *const u64,
                                zeta as *const F as *const u64)
    };
    if rc != 0 {
        warn_once("zkhip_coeff_to_extended", rc);
    }
    rc == 0
}

/// `EvaluationDomain::extended_to_coeff` for F = bn256::Fr; `a` (2^ext_k evaluations) is consumed, `out` receives out.len() coefficients.
pub(crate) fn try_extended_to_coeff<F: 'static>(a: &mut [F], ext_k: u32, ext_omega_inv: &F, ext_divisor: &F, zeta: &F, out: &mut [F]) -> bool {
    if !is::<F, Fr>() || ext_k > 28 || a.len() != 1usize << ext_k || out.len() > a.len() || !usable() {
        return false;
    }
    // SAFETY: sizes checked above; distinct borrows
    let rc = unsafe {
        zkhip_extended_to_coeff(a.as_mut_ptr() as *mut u64, ext_k, ext_omega_inv as *const F as *const u64, ext_divisor as *const F as *const u64,
                                zeta as *const F as *const u64, out.as_mut_ptr() as *mut u64, out.len())
    };
    if rc != 0 {
        warn_once("zkhip_extended_to_coeff", rc);
    }
    rc == 0
}

/// `EvaluationDomain::divide_by_vanishing_poly` for F = bn256::Fr: a[i] *= table[i % table.len()] (table = the inverted t_evaluations).
pub(crate) fn try_mul_periodic<F: 'static>(a: &mut [F], table: &[F]) -> bool {
    if !is::<F, Fr>() || table.is_empty() || table.len() > u32::MAX as usize || !usable() {
        return false;
    }
    // SAFETY: `a` in place, `table` read only
    let rc = unsafe { zkhip_mul_periodic(a.as_mut_ptr() as *mut u64, a.len(), table.as_ptr() as *const u64, table.len() as u32) };
    if rc != 0 {
        warn_once("zkhip_mul_periodic", rc);
    }
    rc == 0
}


// ======================================================================================================================================
// prover_patch.rs, mode (a): all columns of one phase through ONE call (the library shares launch sets between the vectors of a batch:
// 256 commits of 2^13 points take 6.0 ms instead of 68 ms one by one -- bench.py `small_circuit_replays`)
// ======================================================================================================================================

/// Columns above this many elements gain nothing from sharing a launch set (each MSM / transform fills the card alone) and the gather
/// into one contiguous buffer would cost a host copy of 32 n bytes per column: they stay on the per-column path.
const MAX_BATCHED_LOG: u32 = 18;

/// `params.commit_lagrange` / `params.commit` of every column of a phase: Some(results in column order) when C = bn256::G1Affine and the
/// batched call succeeded, None otherwise (the caller runs its per-column loop).  Every column must have `bases.len()` or fewer elements
/// and all columns the same length.
pub(crate) fn try_msm_g1_many<C: 'static, S: 'static, P: 'static + Clone>(columns: &[&[S]], bases: &[C], identity: P) -> Option<Vec<P>> {
    let batch = columns.len();
    let n = columns.first().map_or(0, |c| c.len());
    if !(is::<C, G1Affine>() && is::<S, Fr>() && is::<P, G1>()) || batch < 2 || n < MIN_GPU_MSM || n > bases.len() || n > 1usize << MAX_BATCHED_LOG
        || columns.iter().any(|c| c.len() != n) || !usable() {
        return None;
    }
    let mut flat: Vec<u64> = Vec::with_capacity(batch * n * 4);
    for c in columns {
        // SAFETY: &[S] = &[Fr]: n elements of 4 little-endian u64 limbs
        flat.extend_from_slice(unsafe { std::slice::from_raw_parts(c.as_ptr() as *const u64, n * 4) });
    }
    let mut out = vec![identity; batch];
    // SAFETY: batch contiguous vectors of n scalars, bases[..n], batch results of 12 limbs each (P = G1)
    let rc = unsafe { zkhip_msm_g1_batch(flat.as_ptr(), bases.as_ptr() as *const u64, n, batch, out.as_mut_ptr() as *mut u64) };
    if rc != 0 {
        warn_once("zkhip_msm_g1_batch", rc);
        return None;
    }
    Some(out)
}

/// `lagrange_to_coeff` of every column of a phase, in place: true when F = bn256::Fr and the batched call succeeded.
pub(crate) fn try_ifft_scaled_many<F: 'static>(columns: &mut [&mut [F]], omega_inv: &F, log_n: u32, divisor: &F) -> bool {
    let batch = columns.len();
    let n = 1usize << log_n.min(28);
    if !is::<F, Fr>() || batch < 2 || log_n < MIN_GPU_NTT_LOG || log_n > MAX_BATCHED_LOG || columns.iter().any(|c| c.len() != n) || !usable() {
        return false;
    }
    let mut flat: Vec<u64> = Vec::with_capacity(batch * n * 4);
    for c in columns.iter() {
        // SAFETY: &[F] = &[Fr]
        flat.extend_from_slice(unsafe { std::slice::from_raw_parts(c.as_ptr() as *const u64, n * 4) });
    }
    // SAFETY: batch contiguous polynomials of 2^log_n elements, transformed in place
    let rc = unsafe { zkhip_ifft_scaled_batch(flat.as_mut_ptr(), omega_inv as *const F as *const u64, log_n, divisor as *const F as *const u64, batch as u32) };
    if rc != 0 {
        warn_once("zkhip_ifft_scaled_batch", rc);       // the caller's columns are untouched: its per-column loop still sees the inputs
        return false;
    }
    for (i, c) in columns.iter_mut().enumerate() {
        // SAFETY: as above, mutable
        unsafe { std::slice::from_raw_parts_mut(c.as_mut_ptr() as *mut u64, n * 4) }.copy_from_slice(&flat[i * n * 4..(i + 1) * n * 4]);
    }
    true
}

/// `coeff_to_extended` of every column of the quotient phase: Some(extended evaluations, one Vec per column) or None.
pub(crate) fn try_coeff_to_extended_many<F: 'static + Clone>(columns: &[&[F]], k: u32, ext_k: u32, ext_omega: &F, zeta: &F, zero: F) -> Option<Vec<Vec<F>>> {
    let batch = columns.len();
    let (n, en) = (1usize << k.min(28), 1usize << ext_k.min(28));
    if !is::<F, Fr>() || batch < 2 || ext_k > MAX_BATCHED_LOG || k > ext_k || columns.iter().any(|c| c.len() != n) || !usable() {
        return None;
    }
    let mut flat: Vec<u64> = Vec::with_capacity(batch * n * 4);
    for c in columns {
        // SAFETY: &[F] = &[Fr]
        flat.extend_from_slice(unsafe { std::slice::from_raw_parts(c.as_ptr() as *const u64, n * 4) });
    }
    let mut ext: Vec<u64> = vec![0; batch * en * 4];
    // SAFETY: batch polynomials of 2^k coefficients in, batch x 2^ext_k evaluations out, distinct buffers
    let rc = unsafe { zkhip_coeff_to_extended_batch(flat.as_ptr(), k, ext.as_mut_ptr(), ext_k, batch as u32, ext_omega as *const F as *const u64,
                                                    zeta as *const F as *const u64) };
    if rc != 0 {
        warn_once("zkhip_coeff_to_extended_batch", rc);
        return None;
    }
    Some((0..batch).map(|i| {
        let mut v = vec![zero.clone(); en];
        // SAFETY: Vec<F> = Vec<Fr> of en elements
        unsafe { std::slice::from_raw_parts_mut(v.as_mut_ptr() as *mut u64, en * 4) }.copy_from_slice(&ext[i * en * 4..(i + 1) * en * 4]);
        v
    }).collect())
}

// ======================================================================================================================================
// prover_patch.rs, mode (b): device-resident columns.  A `DevCols` is `count` columns of `len` field elements in ONE zkhip_alloc'd block
// (column i at element offset i * len): what `Vec<Polynomial<Fr, _>>` becomes while a proof is in flight.  Only commitments (96 bytes) and
// evaluations (32 bytes) cross PCIe after the witness has gone up.  Every method returns false / None on a non-zero status; the caller then
// abandons the device-resident proof and re-runs create_proof's CPU-orchestrated body (the witness columns are still on the host).
// ======================================================================================================================================
pub(crate) struct DevCols { ptr: *mut c_void, pub len: usize, pub count: usize }

// SAFETY: the handle is a device address; the library serialises access per stream and create_proof drives one proof from one thread
unsafe impl Send for DevCols {}

impl DevCols {
    pub(crate) fn alloc(len: usize, count: usize) -> Option<DevCols> {
        if !usable() { return None; }
        let mut ptr: *mut c_void = std::ptr::null_mut();
        // SAFETY: out-parameter of the allocation
        let rc = unsafe { zkhip_alloc(len * count * 32, &mut ptr) };
        if rc != 0 { warn_once("zkhip_alloc", rc); return None; }
        Some(DevCols { ptr, len, count })
    }
    /// device address of element `row` of column `col`
    pub(crate) fn at(&self, col: usize, row: usize) -> *mut c_void {
        debug_assert!(col < self.count && row <= self.len);
        (self.ptr as usize + (col * self.len + row) * 32) as *mut c_void
    }
    /// host slice -> rows [row, row + src.len()) of column `col` (the witness going up; the blinding rows of a product)
    pub(crate) fn upload<F: 'static>(&self, col: usize, row: usize, src: &[F]) -> bool {
        if !is::<F, Fr>() || row + src.len() > self.len { return false; }
        // SAFETY: src = &[Fr]; the destination range was checked
        let rc = unsafe { zkhip_upload(self.at(col, row), src.as_ptr() as *const c_void, src.len() * 32) };
        if rc != 0 { warn_once("zkhip_upload", rc); }
        rc == 0
    }
    pub(crate) fn download<F: 'static>(&self, col: usize, row: usize, dst: &mut [F]) -> bool {
        if !is::<F, Fr>() || row + dst.len() > self.len { return false; }
        // SAFETY: dst = &mut [Fr]
        let rc = unsafe { zkhip_download(dst.as_mut_ptr() as *mut c_void, self.at(col, row), dst.len() * 32) };
        if rc != 0 { warn_once("zkhip_download", rc); }
        rc == 0
    }
    /// commitments of columns [first, first + batch) against a pinned base array (`params.commit_lagrange` / `commit` of a whole phase)
    pub(crate) fn commit_many<C: 'static, P: 'static + Clone>(&self, first: usize, batch: usize, n: usize, bases: &[C], identity: P) -> Option<Vec<P>> {
        if !(is::<C, G1Affine>() && is::<P, G1>()) || first + batch > self.count || n > self.len || n > bases.len() { return None; }
        let res = DevCols::alloc(3, batch)?;                 // batch x 96 bytes
        let mut out = vec![identity; batch];
        // SAFETY: device-resident scalars (column stride = self.len elements), host pointer into a registered array, batch x 12 limbs out
        let rc = unsafe { zkhip_msm_g1_registered_batch_device(bases.as_ptr() as *const u64, self.at(first, 0), n, batch, self.len, res.ptr, std::ptr::null_mut()) };
        // SAFETY: out = Vec<G1> of `batch` elements
        let rc2 = if rc == 0 { unsafe { zkhip_download(out.as_mut_ptr() as *mut c_void, res.ptr, batch * 96) } } else { rc };
        if rc2 != 0 { warn_once("zkhip_msm_g1_registered_batch_device", rc2); return None; }
        Some(out)
    }
    /// `lagrange_to_coeff` of columns [first, first + batch) in place
    pub(crate) fn ifft_scaled_many<F: 'static>(&self, first: usize, batch: usize, omega_inv: &F, log_n: u32, divisor: &F) -> bool {
        if !is::<F, Fr>() || first + batch > self.count || self.len < 1usize << log_n.min(28) { return false; }
        // SAFETY: in place on device memory this handle owns
        let rc = unsafe { zkhip_ifft_scaled_batch_device(self.at(first, 0), omega_inv as *const F as *const u64, log_n, divisor as *const F as *const u64,
                                                         batch as u32, self.len, std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_ifft_scaled_batch_device", rc); }
        rc == 0
    }
    /// `coeff_to_extended` of columns [first, first + batch) into columns [out_first, ..) of `out` (len = 2^ext_k)
    pub(crate) fn coeff_to_extended_many<F: 'static>(&self, first: usize, batch: usize, k: u32, out: &DevCols, out_first: usize, ext_k: u32, ext_omega: &F,
                                                     zeta: &F) -> bool {
        if !is::<F, Fr>() || first + batch > self.count || out_first + batch > out.count || out.len != 1usize << ext_k.min(28) { return false; }
        // SAFETY: distinct device blocks
        let rc = unsafe { zkhip_coeff_to_extended_device(self.at(first, 0), self.len, k, out.at(out_first, 0), out.len, ext_k, batch as u32,
                                                         ext_omega as *const F as *const u64, zeta as *const F as *const u64, std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_coeff_to_extended_device", rc); }
        rc == 0
    }
    /// one row program over device-resident columns: `out_col` of `out` = program(columns) for 2^log_rows rows
    pub(crate) fn eval_rows(prog: &VmProgramOwned, columns: &[*const c_void], log_rows: u32, out: &DevCols, out_col: usize) -> bool {
        let p = prog.as_ffi();
        // SAFETY: `p` borrows the Vecs of `prog`, alive for the call; `columns` are device addresses of columns with >= 2^log_rows rows
        let rc = unsafe { zkhip_fr_eval_rows_device(&p, columns.as_ptr(), columns.len() as u32, log_rows, 0, out.at(out_col, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_fr_eval_rows_device", rc); }
        rc == 0
    }
    /// the quotient numerator as a SUM of row programs over consecutive runs of the y-fold's terms (`weights[p]` = y^(number of terms after
    /// run p)): for circuits with hundreds of columns at 2^13 .. 2^15 rows, where ONE program is thousands of instructions walked by a handful
    /// of wavefronts; the programs run side by side in one launch.  `lower_graph` is called once per run of `Calculation`s ending at a `Store`
    pub(crate) fn eval_rows_sum<F: 'static>(progs: &[VmProgramOwned], weights: &[F], columns: &[*const c_void], log_rows: u32, out: &DevCols, out_col: usize) -> bool {
        if !is::<F, Fr>() || progs.is_empty() || progs.len() != weights.len() { return false; }
        let ffi: Vec<VmProgram> = progs.iter().map(|p| p.as_ffi()).collect();
        // SAFETY: `ffi` borrows the Vecs of `progs`, alive for the call; weights = &[Fr]; columns are device addresses of >= 2^log_rows rows
        let rc = unsafe { zkhip_fr_eval_rows_sum_device(ffi.as_ptr(), weights.as_ptr() as *const u64, ffi.len() as u32, columns.as_ptr(), columns.len() as u32, log_rows,
                                                        out.at(out_col, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_fr_eval_rows_sum_device", rc); }
        rc == 0
    }
    /// column `out_col` of `out` = sum_j coeffs[j] * columns[j] over n rows (the linear combinations of lookups' compressed expressions with
    /// many inputs, of the multi-open argument when it is assembled on this side)
    pub(crate) fn linear_combination<F: 'static>(columns: &[*const c_void], coeffs: &[F], n: usize, out: &DevCols, out_col: usize) -> bool {
        if !is::<F, Fr>() || columns.len() != coeffs.len() || n > out.len { return false; }
        // SAFETY: coeffs = &[Fr]; columns are device addresses of >= n rows
        let rc = unsafe { zkhip_fr_linear_combination_device(columns.as_ptr(), coeffs.as_ptr() as *const u64, columns.len(), n, out.at(out_col, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_fr_linear_combination_device", rc); }
        rc == 0
    }
    /// z = grand product of num / den (num = column `z_col` on entry, den is consumed): `permutation::Argument::commit`, `lookup::commit_product`
    pub(crate) fn grand_product(&self, z_col: usize, den: &DevCols, den_col: usize, n: usize) -> bool {
        // SAFETY: z aliases num as the C ABI allows
        let rc = unsafe { zkhip_fr_grand_product_device(self.at(z_col, 0), den.at(den_col, 0), n, self.at(z_col, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_fr_grand_product_device", rc); }
        rc == 0
    }
    /// every set's product column of the permutation argument in ONE call (`permutation::Argument::commit`'s loop over
    /// `columns.chunks(chunk_len)`): `values[c]` / `sigmas[c]` = device addresses of permutation column c and of its sigma column in the
    /// Lagrange basis; columns [z_first, z_first + ceil(columns / chunk_len)) of self receive z, chained through z[usable_rows]; the
    /// caller then uploads its blinding rows behind usable_rows as upstream does
    pub(crate) fn permutation_products<F: 'static>(&self, z_first: usize, values: &[*const c_void], sigmas: &[*const c_void], chunk_len: usize, log_n: u32,
                                                   usable_rows: usize, beta: &F, gamma: &F, delta: &F, omega: &F) -> bool {
        let sets = (values.len() + chunk_len.max(1) - 1) / chunk_len.max(1);
        if !is::<F, Fr>() || values.len() != sigmas.len() || chunk_len == 0 || self.len != 1usize << log_n.min(28) || z_first + sets > self.count { return false; }
        // SAFETY: z columns are adjacent in this block (stride = len = 2^log_n elements: the dense [sets][n] layout the C ABI writes)
        let rc = unsafe { zkhip_permutation_products_device(values.as_ptr(), sigmas.as_ptr(), values.len() as u32, chunk_len as u32, log_n, usable_rows,
                                                            beta as *const F as *const u64, gamma as *const F as *const u64, delta as *const F as *const u64,
                                                            omega as *const F as *const u64, self.at(z_first, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_permutation_products_device", rc); }
        rc == 0
    }
    /// `permute_expression_pair`: columns (input, table) of self -> (permuted_input, permuted_table) of `out`, first `usable_rows` rows
    pub(crate) fn lookup_permute(&self, input: usize, table: usize, usable_rows: usize, out: &DevCols, pin: usize, ptab: usize) -> bool {
        // SAFETY: outputs do not alias the inputs (different blocks or different columns)
        let rc = unsafe { zkhip_lookup_permute_device(self.at(input, 0), self.at(table, 0), usable_rows, out.at(pin, 0), out.at(ptab, 0), std::ptr::null_mut()) };
        if rc != 0 { warn_once("zkhip_lookup_permute_device", rc); }   // ZKHIP_EINVAL = an input value missing from the table (Error::ConstraintSystemFailure upstream)
        rc == 0
    }
    /// h(X) (X^n - 1) evaluations in column `col` -> divided by the vanishing polynomial -> `out_len` coefficients in column `out_col` of `out`
    pub(crate) fn quotient_to_coeff<F: 'static>(&self, col: usize, ext_k: u32, t_inv: &DevCols, period: u32, ext_omega_inv: &F, ext_divisor: &F, zeta: &F,
                                                out: &DevCols, out_col: usize, out_len: usize) -> bool {
        if !is::<F, Fr>() || self.len != 1usize << ext_k.min(28) || out_len > out.len { return false; }
        // SAFETY: in place, then into a distinct block
        let rc = unsafe { zkhip_mul_periodic_device(self.at(col, 0), self.len, t_inv.at(0, 0), period, std::ptr::null_mut()) };
        let rc = if rc != 0 { rc } else { unsafe {
            zkhip_extended_to_coeff_device(self.at(col, 0), self.len, ext_k, ext_omega_inv as *const F as *const u64, ext_divisor as *const F as *const u64,
                                           zeta as *const F as *const u64, out.at(out_col, 0), out.len, out_len, 1, std::ptr::null_mut()) } };
        if rc != 0 { warn_once("zkhip_extended_to_coeff_device", rc); }
        rc == 0
    }
    /// evaluations of `polys` (device addresses of n-coefficient polynomials) at `point`: the `eval_polynomial` loop of create_proof in one call
    pub(crate) fn eval_polys<F: 'static + Clone>(polys: &[*const c_void], n: usize, point: &F, zero: F) -> Option<Vec<F>> {
        if !is::<F, Fr>() { return None; }
        let res = DevCols::alloc(polys.len().max(1), 1)?;
        let mut out = vec![zero; polys.len()];
        // SAFETY: polys.len() results of 32 bytes each
        let rc = unsafe { zkhip_fr_eval_polynomial_batch_device(polys.as_ptr(), polys.len(), n, point as *const F as *const u64, res.ptr, std::ptr::null_mut()) };
        let rc2 = if rc == 0 { unsafe { zkhip_download(out.as_mut_ptr() as *mut c_void, res.ptr, polys.len() * 32) } } else { rc };
        if rc2 != 0 { warn_once("zkhip_fr_eval_polynomial_batch_device", rc2); return None; }
        Some(out)
    }
}

impl Drop for DevCols {
    fn drop(&mut self) {
        // SAFETY: allocated by zkhip_alloc, freed once (zkhip_free waits for queued work on the block)
        let _ = unsafe { zkhip_free(self.ptr) };
    }
}

/// One query of the multi-open argument over device-resident polynomials: what `ProverQuery { point, poly, blind }` becomes when `poly`
/// lives in a `DevCols` (prover_patch.rs mode (b)); `eval` as written to the transcript.
pub(crate) fn dev_query<F: 'static>(point: &F, d_poly: *const c_void, eval: &F) -> Option<ProverQueryC> {
    if !is::<F, Fr>() { return None; }
    // SAFETY: F = Fr = 4 x u64 (checked)
    let (p, e) = unsafe { (*(point as *const F as *const [u64; 4]), *(eval as *const F as *const [u64; 4])) };
    Some(ProverQueryC { point: p, d_poly, eval: e, has_eval: 1, reserved: 0 })
}

/// `ProverGWC::create_proof` on device-resident polynomials: one witness commitment per distinct point (order of first appearance),
/// committed against the pinned `params.g`.  None = not taken (the caller downloads the polynomials and runs the upstream prover).
pub(crate) fn multiopen_gwc<C: 'static, F: 'static, P: 'static + Clone>(g: &[C], k: u32, queries: &[ProverQueryC], v: &F, identity: P) -> Option<Vec<P>> {
    if !(is::<C, G1Affine>() && is::<F, Fr>() && is::<P, G1>()) || !usable() || queries.is_empty() || g.len() < 1usize << k.min(28) { return None; }
    let mut out = vec![identity; queries.len()];           // at most one witness per query
    let mut n_out = 0usize;
    // SAFETY: `g` is a registered base array (Pinned<C>), queries are repr(C), out has room for queries.len() points of 12 limbs
    let rc = unsafe { zkhip_multiopen_gwc_device(g.as_ptr() as *const u64, k, queries.as_ptr(), queries.len(), v as *const F as *const u64,
                                                 out.as_mut_ptr() as *mut u64, out.len(), &mut n_out) };
    if rc != 0 { warn_once("zkhip_multiopen_gwc_device", rc); return None; }
    out.truncate(n_out);
    Some(out)
}

/// `ProverSHPLONK::create_proof` in the two steps the transcript imposes: `begin` (y, v squeezed) -> H; the caller writes H and squeezes u;
/// `finish` -> H'.  Dropping a session that was not finished releases the library's state.
pub(crate) struct ShplonkSession { state: *mut ShplonkState }
impl ShplonkSession {
    pub(crate) fn begin<C: 'static, F: 'static, P: 'static + Clone>(g: &[C], k: u32, queries: &[ProverQueryC], y: &F, v: &F, identity: P) -> Option<(ShplonkSession, P)> {
        if !(is::<C, G1Affine>() && is::<F, Fr>() && is::<P, G1>()) || !usable() || queries.is_empty() || g.len() < 1usize << k.min(28) { return None; }
        let mut h = identity;
        let mut state: *mut ShplonkState = std::ptr::null_mut();
        // SAFETY: as multiopen_gwc; `h` = G1 = 12 limbs; `state` is an out-parameter
        let rc = unsafe { zkhip_multiopen_shplonk_begin_device(g.as_ptr() as *const u64, k, queries.as_ptr(), queries.len(), y as *const F as *const u64,
                                                               v as *const F as *const u64, &mut h as *mut P as *mut u64, &mut state) };
        if rc != 0 || state.is_null() { warn_once("zkhip_multiopen_shplonk_begin_device", rc); return None; }
        Some((ShplonkSession { state }, h))
    }
    pub(crate) fn finish<F: 'static, P: 'static + Clone>(mut self, u: &F, identity: P) -> Option<P> {
        if !(is::<F, Fr>() && is::<P, G1>()) { return None; }        // (self drops: the state is released by Drop)
        let mut hp = identity;
        let state = std::mem::replace(&mut self.state, std::ptr::null_mut());   // finish releases the state whatever it returns
        // SAFETY: `state` came from begin and is used once
        let rc = unsafe { zkhip_multiopen_shplonk_finish_device(state, u as *const F as *const u64, &mut hp as *mut P as *mut u64) };
        if rc != 0 { warn_once("zkhip_multiopen_shplonk_finish_device", rc); return None; }
        Some(hp)
    }
}
impl Drop for ShplonkSession {
    fn drop(&mut self) {
        if !self.state.is_null() {
            // SAFETY: a state that finish never consumed
            let _ = unsafe { zkhip_multiopen_shplonk_abort(self.state) };
        }
    }
}

/// A row program with its storage: what evaluation.rs builds from a `GraphEvaluator` (prover_patch.rs `lower_graph`).
pub(crate) struct VmProgramOwned {
    pub insns: Vec<VmInsn>,
    pub constants: Vec<[u64; 4]>,        // Montgomery limbs of bn256::Fr
    pub rotations: Vec<i32>,
    pub rot_scale: i32,
    pub result_reg: u32,
    pub omega: Option<[u64; 4]>,
}

impl VmProgramOwned {
    fn as_ffi(&self) -> VmProgram {
        VmProgram {
            insns: self.insns.as_ptr(), n_insns: self.insns.len() as u32,
            constants: self.constants.as_ptr() as *const u64, n_constants: self.constants.len() as u32,
            rotations: self.rotations.as_ptr(), n_rotations: self.rotations.len() as u32,
            rot_scale: self.rot_scale, result_reg: self.result_reg,
            omega: self.omega.as_ref().map_or(std::ptr::null(), |w| w.as_ptr()),
        }
    }
}
