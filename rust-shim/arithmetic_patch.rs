//! Edits to `halo2_proofs/src/arithmetic.rs` of halo2-axiom [DEP; the crate the reference imports as `halo2_base::halo2_proofs`,
//! /root/reference/aggregator/Cargo.toml:7-8].  The two public functions keep their upstream GENERIC signatures -- `ParamsKZG<E: Engine>::commit`,
//! `g_to_lagrange` and `EvaluationDomain<F>` call them generically, so a monomorphic replacement would not type-check -- and gain a
//! three-line prologue each; the upstream bodies stay, renamed `*_cpu`, as the path of every other instantiation (`best_multiexp::<G2Affine>`,
//! `best_fft::<_, G1>` in `g_to_lagrange`, pasta / secp curves) and the fall-back when the library reports a non-zero status.
//!
//! How to apply: (1) rename the existing `pub fn best_multiexp` to `fn best_multiexp_cpu` and `pub fn best_fft` to `fn best_fft_cpu`
//! (bodies unchanged); (2) add the two functions below.  Nothing else in the file changes.

use crate::zkhip_ffi;

/// Performs a multi-exponentiation operation: sum_i coeffs[i] * bases[i].  Panics if coeffs and bases have a different length
/// (upstream contract, unchanged).  BN254 G1 runs on the GPU (`zkhip_msm_g1`, include/zkhip.h); a `bases` slice inside an array pinned
/// by `ParamsKZG` (commitment_patch.rs) takes the fixed-base path without any upload of points.
pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {
    assert_eq!(coeffs.len(), bases.len());
    if let Some(acc) = zkhip_ffi::try_msm_g1::<C, C::Scalar, C::Curve>(coeffs, bases, C::Curve::identity()) {
        return acc;
    }
    best_multiexp_cpu(coeffs, bases)
}

/// Performs a radix-2 Fast-Fourier Transformation on a vector of size n = 2^log_n, in place, natural order in and out (upstream
/// contract, unchanged).  `G = Scalar = bn256::Fr` runs on the GPU (`zkhip_ntt_fr`); the curve-point instantiation used by
/// `g_to_lagrange` and every other field stay on the crate's CPU code.
pub fn best_fft<Scalar: Field, G: FftGroup<Scalar>>(a: &mut [G], omega: Scalar, log_n: u32) {
    if zkhip_ffi::try_ntt_fr::<Scalar, G>(a, &omega, log_n) {
        return;
    }
    best_fft_cpu(a, omega, log_n)
}
