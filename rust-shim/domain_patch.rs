//! Optional edits to `halo2_proofs/src/poly/domain.rs` of halo2-axiom [DEP]: with arithmetic_patch.rs alone every `EvaluationDomain` method
//! already reaches the GPU through `best_fft`, but pays separate CPU passes for the scale / zeta-power / zero-padding loops around it.
//! The four prologues below hand the whole method to one fused call each; the upstream body stays below the prologue as the path of
//! every field but bn256::Fr and the fall-back on a non-zero status.  `F: 'static` holds through `ff::Field`.
//!
//! Field names as upstream: `self.k`, `self.extended_k`, `self.extended_omega`, `self.extended_omega_inv`, `self.extended_ifft_divisor`,
//! `self.g_coset` (= F::ZETA), `self.g_coset_inv` (= zeta^2), `self.t_evaluations`, `self.quotient_poly_degree`, `self.n`.

use crate::zkhip_ffi;

impl<F: WithSmallOrderMulGroup<3>> EvaluationDomain<F> {
    // fn ifft(a: &mut [F], omega_inv: F, log_n: u32, divisor: F) -- first statement:
    //
    //     if zkhip_ffi::try_ifft_scaled::<F>(a, &omega_inv, log_n, &divisor) { return; }

    // pub fn coeff_to_extended(&self, p: &Polynomial<F, Coeff>) -> Polynomial<F, ExtendedLagrangeCoeff> -- first statements:
    //
    //     assert_eq!(p.values.len(), 1 << self.k);
    //     {
    //         let mut out = vec![F::ZERO; self.extended_len()];
    //         if zkhip_ffi::try_coeff_to_extended::<F>(&p.values, self.k, &mut out, self.extended_k, &self.extended_omega, &self.g_coset) {
    //             return Polynomial { values: out, _marker: PhantomData };
    //         }
    //     }

    // pub fn extended_to_coeff(&self, mut a: Polynomial<F, ExtendedLagrangeCoeff>) -> Vec<F> -- first statements:
    //
    //     assert_eq!(a.values.len(), self.extended_len());
    //     {
    //         let mut out = vec![F::ZERO; (self.n * self.quotient_poly_degree) as usize];
    //         if zkhip_ffi::try_extended_to_coeff::<F>(&mut a.values, self.extended_k, &self.extended_omega_inv, &self.extended_ifft_divisor,
    //                                                  &self.g_coset, &mut out) {
    //             return out;
    //         }
    //     }
    //   (the library derives zeta^-1 = zeta^2 from `zeta` itself, as `EvaluationDomain::new` does for `g_coset_inv`)

    // pub fn divide_by_vanishing_poly(&self, mut a: Polynomial<F, ExtendedLagrangeCoeff>) -> Polynomial<F, ExtendedLagrangeCoeff> -- first statements:
    //
    //     assert_eq!(a.values.len(), self.extended_len());
    //     if zkhip_ffi::try_mul_periodic::<F>(&mut a.values, &self.t_evaluations) { return Polynomial { values: a.values, _marker: PhantomData }; }
}
