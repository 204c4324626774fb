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
use std::os::raw::{c_char, c_int};
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
}

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

/// `ParamsKZG::{setup, read, read_custom, from_parts, clone, downsize}`: pin `g` / `g_lagrange` on the device(s) and build the fixed-base
/// tables, so that `commit` / `commit_lagrange` (which pass `&self.g[..poly.len()]`) upload 32 n bytes of scalars and nothing else.
/// No-op for every curve but bn256::G1Affine.  The memory must stay alive and unchanged until `unpin` (the `Drop` impl in commitment_patch.rs).
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

/// Undo `pin`; must run before the memory is freed or rewritten (`Drop for ParamsKZG`, `downsize`).
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
