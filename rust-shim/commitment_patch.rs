//! Edits to `halo2_proofs/src/poly/kzg/commitment.rs` of halo2-axiom [DEP]: residency of `ParamsKZG::{g, g_lagrange}` on the GPU(s).
//! The SRS is static per params object (`gen_srs`, /root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69; `ParamsKZG::setup`,
//! /root/reference/voter/benches/voter_circuit.rs:60), so it is uploaded and turned into fixed-base tables once: every constructor pins
//! both arrays, `Drop` unpins them BEFORE the Vecs are freed -- an allocator that hands the same address range to another SRS must never
//! meet a stale registration (the library also keeps sampled points as a second guard, but this Drop impl is the contract).
//! `commit` / `commit_lagrange` are untouched: they call `best_multiexp(&scalars, &self.g[..size])` and `zkhip_msm_g1` recognises any
//! sub-range of a pinned array by address.  All of this is a no-op unless `E::G1Affine` is bn256::G1Affine (zkhip_ffi::pin checks the TypeId).
//!
//! How to apply:
//!   1. remove `Clone` from the `#[derive(Debug, Clone)]` of `pub struct ParamsKZG<E: Engine>` (the manual impl below pins the copy);
//!   2. add the three impl blocks below;
//!   3. end every constructor with `.zkhip_pinned()`:
//!        `setup`        :  `Self { k, n, g, g_lagrange, g2, s_g2 }.zkhip_pinned()`
//!        `from_parts`   :  `Self { k, n: 1 << k, g_lagrange: ..., g, g2, s_g2 }.zkhip_pinned()`
//!        `read_custom`  :  `Ok(Self { k, n: n as u64, g, g_lagrange, g2, s_g2 }.zkhip_pinned())`     (`read` forwards to it)
//!   4. `Params::downsize` (it truncates `g` and replaces `g_lagrange`): first statement `self.zkhip_unpin();`, last statement `self.zkhip_pin();`.
//!   5. `g_to_lagrange<C: CurveAffine>(g_projective: Vec<C::Curve>, k: u32) -> Vec<C>` -- first statement:
//!        `if let Some(v) = zkhip_ffi::try_g_to_lagrange::<C, C::Curve>(&g_projective, k, C::identity()) { return v; }`
//!      (the upstream body -- best_fft over the points, `*g *= n_inv`, batch_normalize -- stays below it for every other curve and as the fall-back).
//! A struct with a `Drop` impl cannot be destructured by move; the crate never does that with `ParamsKZG` (it is only read through
//! `&self`: `get_g`, `g2`, `s_g2`, `commit*`, `verifier_params`).

use crate::zkhip_ffi;

impl<E: Engine> ParamsKZG<E> {
    /// Pin `g` and `g_lagrange` (moving `self` afterwards does not move the Vecs' heap memory).
    fn zkhip_pin(&self) {
        zkhip_ffi::pin::<E::G1Affine>(&self.g);
        zkhip_ffi::pin::<E::G1Affine>(&self.g_lagrange);
    }

    fn zkhip_unpin(&self) {
        zkhip_ffi::unpin::<E::G1Affine>(&self.g);
        zkhip_ffi::unpin::<E::G1Affine>(&self.g_lagrange);
    }

    fn zkhip_pinned(self) -> Self {
        self.zkhip_pin();
        self
    }
}

impl<E: Engine> Drop for ParamsKZG<E> {
    fn drop(&mut self) {
        self.zkhip_unpin();       // runs before the fields (the Vecs) are dropped
    }
}

impl<E: Engine> Clone for ParamsKZG<E> {
    fn clone(&self) -> Self {
        Self {
            k: self.k,
            n: self.n,
            g: self.g.clone(),
            g_lagrange: self.g_lagrange.clone(),
            g2: self.g2,
            s_g2: self.s_g2,
        }
        .zkhip_pinned()
    }
}
