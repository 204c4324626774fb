//! Edits to `halo2_proofs/src/poly/kzg/commitment.rs` of halo2-axiom [DEP]: residency of `ParamsKZG::{g, g_lagrange}` on the GPU(s).
//! The SRS is static per params object (`gen_srs`, /root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69; `ParamsKZG::setup`,
//! /root/reference/voter/benches/voter_circuit.rs:60), so it is uploaded and turned into fixed-base tables once: both arrays live in a
//! `Pinned<C>` -- a `Vec<C>` that registers its heap block when it is made and unregisters it in ITS OWN `Drop`, before the block is freed
//! (an allocator that hands the same address range to another SRS must never meet a stale registration; the library also keeps sampled
//! points as a second guard, but this Drop is the contract).  `ParamsKZG` itself gains NO `Drop` impl and keeps its `#[derive(Clone)]`
//! (round 4 put `Drop` on `ParamsKZG`, which makes any move-destructuring or struct-update of it anywhere in the dependency tree a hard
//! error E0509; a field type with its own Drop has no such effect).
//! `commit` / `commit_lagrange` are untouched: they call `best_multiexp(&scalars, &self.g[..size])` and `zkhip_msm_g1` recognises any
//! sub-range of a pinned array by address.  All of this is a no-op unless `C` is bn256::G1Affine (zkhip_ffi::pin checks the TypeId).
//!
//! Cost to know about: every pinned array holds a fixed-base table on the device (64 B x windows per point: 3.4 GiB for 2^22 points; arrays
//! of <= 2^15 points also a direct table of 256 KiB per point, within the library's $ZKHIP_DIRECT_BUDGET), and `ParamsKZG::clone()` pins
//! the copy as well -- a host that clones its params per proof should share them behind an `Arc` instead.
//!
//! How to apply:
//!   1. add the `Pinned` type below (top of the file);
//!   2. in `pub struct ParamsKZG<E: Engine>` change the two field types: `pub(crate) g: Pinned<E::G1Affine>`, `pub(crate) g_lagrange: Pinned<E::G1Affine>`
//!      (every read in the crate goes through `&self.g[..]`, `self.g.len()`, `self.g.iter()`: unchanged by `Deref<Target = [C]>`);
//!   3. wrap the Vecs where the struct is built:
//!        `setup`        :  `Self { k, n, g: Pinned::new(g), g_lagrange: Pinned::new(g_lagrange), g2, s_g2 }`
//!        `from_parts`   :  `Self { k, n: 1 << k, g_lagrange: Pinned::new(...), g: Pinned::new(g), g2, s_g2 }`
//!        `read_custom`  :  `Ok(Self { k, n: n as u64, g: Pinned::new(g), g_lagrange: Pinned::new(g_lagrange), g2, s_g2 })`     (`read` forwards to it)
//!   4. `Params::downsize` (it truncates `g` and replaces `g_lagrange`): `self.g.truncate(self.n as usize)` is `Pinned::truncate`;
//!      `self.g_lagrange = Pinned::new(g_to_lagrange(..))` drops (= unpins) the old array;
//!   5. `write_custom` / `write` iterate `self.g.iter()` / `self.g_lagrange.iter()`: unchanged;
//!   6. `g_to_lagrange<C: CurveAffine>(g_projective: Vec<C::Curve>, k: u32) -> Vec<C>` -- first statement:
//!        `if let Some(v) = zkhip_ffi::try_g_to_lagrange::<C, C::Curve>(&g_projective, k, C::identity()) { return v; }`
//!      (the upstream body -- best_fft over the points, `*g *= n_inv`, batch_normalize -- stays below it for every other curve and as the fall-back).

use crate::zkhip_ffi;

/// A `Vec<C>` that is registered with libzkhip for as long as it lives (read-only: the table on the device is built from the contents
/// at registration).  Moving a `Pinned` moves the Vec's header, not its heap block: the registration stays valid.
pub(crate) struct Pinned<C: 'static>(Vec<C>);

impl<C: 'static> Pinned<C> {
    pub(crate) fn new(v: Vec<C>) -> Self {
        zkhip_ffi::pin::<C>(&v);
        Pinned(v)
    }

    /// `Vec::truncate` under a registration: the table is rebuilt for the shorter array (`Params::downsize`)
    pub(crate) fn truncate(&mut self, len: usize) {
        zkhip_ffi::unpin::<C>(&self.0);
        self.0.truncate(len);
        zkhip_ffi::pin::<C>(&self.0);
    }
}

impl<C: 'static> Drop for Pinned<C> {
    fn drop(&mut self) {
        zkhip_ffi::unpin::<C>(&self.0);       // runs before the Vec (field 0) is dropped
    }
}

impl<C: 'static + Clone> Clone for Pinned<C> {
    fn clone(&self) -> Self {
        Pinned::new(self.0.clone())
    }
}

impl<C: 'static> std::ops::Deref for Pinned<C> {
    type Target = [C];
    fn deref(&self) -> &[C] {
        &self.0
    }
}

impl<C: 'static + std::fmt::Debug> std::fmt::Debug for Pinned<C> {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        self.0.fmt(f)
    }
}
