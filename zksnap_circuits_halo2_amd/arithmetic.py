"""`halo2_proofs::arithmetic` mirror: `best_multiexp`, `best_fft` [DEP halo2-axiom] -- the two free functions
the reference reaches through `create_proof` (/root/reference/aggregator/src/wrapper.rs:129).  Same names and
argument meaning; arrays are numpy uint64 views of the Rust memory formats."""
from __future__ import annotations

import numpy as np

from . import _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data


def best_multiexp(coeffs: np.ndarray, bases: np.ndarray) -> np.ndarray:
    """sum_i coeffs[i] * bases[i].  coeffs: (n,4) uint64 Fr, bases: (n,8) uint64 G1Affine -> (12,) uint64 G1."""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    if not (isinstance(bases, np.ndarray) and bases.dtype == np.uint64 and bases.flags.c_contiguous):
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
    bases2 = bases.reshape(-1, 8)
    # same contract as the reference: `assert_eq!(coeffs.len(), bases.len())`
    assert coeffs.shape[0] == bases2.shape[0], "coeffs.len() != bases.len()"
    out = np.zeros(12, dtype=np.uint64)
    lib = _lib.load()
    _lib.check(lib.zkhip_msm_g1(_ptr(coeffs), _ptr(bases2), coeffs.shape[0], _ptr(out)))
    return out


def best_multiexp_g2(coeffs: np.ndarray, bases: np.ndarray) -> np.ndarray:
    """`best_multiexp::<G2Affine>`: coeffs (n,4) uint64 Fr, bases (n,16) uint64 G2Affine -> (24,) uint64 G2 (Jacobian over Fq2)."""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    bases2 = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 16)
    assert coeffs.shape[0] == bases2.shape[0], "coeffs.len() != bases.len()"
    out = np.zeros(24, dtype=np.uint64)
    _lib.check(_lib.load().zkhip_msm_g2(_ptr(coeffs), _ptr(bases2), coeffs.shape[0], _ptr(out)))
    return out


def best_fft(a: np.ndarray, omega: np.ndarray, log_n: int) -> None:
    """In-place NTT of a ((2^log_n, 4) uint64 Fr) with the (4,) uint64 root `omega`."""
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    assert a.size == 4 << log_n, "a.len() != 1 << log_n"
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    lib = _lib.load()
    _lib.check(lib.zkhip_ntt_fr(_ptr(a), _ptr(omega), log_n))


def eval_polynomial(poly: np.ndarray, point: np.ndarray) -> np.ndarray:
    """sum poly[i] * point^i  ((n,4) uint64 Fr, (4,) uint64 Fr) -> (4,) uint64 Fr."""
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    point = np.ascontiguousarray(point, dtype=np.uint64).reshape(4)
    out = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.load().zkhip_fr_eval_polynomial(_ptr(poly), poly.shape[0], _ptr(point), _ptr(out)))
    return out


def kate_division(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Quotient of a(X) by (X - b), remainder dropped: (n,4) -> (n-1,4)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(4)
    q = np.zeros((max(a.shape[0] - 1, 0), 4), dtype=np.uint64)
    _lib.check(_lib.load().zkhip_fr_kate_division(_ptr(a), a.shape[0], _ptr(b), _ptr(q)))
    return q


def batch_invert(a: np.ndarray) -> None:
    """In-place elementwise inversion; zeros stay zero (ff::BatchInvert)."""
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    _lib.check(_lib.load().zkhip_fr_batch_invert(_ptr(a), a.size // 4))


def prefix_product(v: np.ndarray) -> np.ndarray:
    """Grand-product running product: out[0] = 1, out[i] = v[0] * ... * v[i-1]."""
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(v)
    _lib.check(_lib.load().zkhip_fr_prefix_product(_ptr(v), v.shape[0], _ptr(out)))
    return out
