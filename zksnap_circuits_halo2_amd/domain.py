"""`halo2_proofs::poly::EvaluationDomain` mirror [DEP halo2-axiom poly/domain.rs]: the
constructor derives the domain constants on the host exactly as `EvaluationDomain::new(j, k)` does, the
transforms run on the GPU through the fused C-ABI entry points."""
from __future__ import annotations

import numpy as np

from . import _lib
from .fields import R_MOD, ZETA, fr_encode, omega_for


class EvaluationDomain:
    def __init__(self, j: int, k: int, zeta: int = ZETA):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        ext_omega = omega_for(ek)
        omega = pow(ext_omega, 1 << (ek - k), R_MOD)
        self._omega, self._ext_omega = omega, ext_omega
        enc = lambda v: fr_encode([v])[0]
        self.omega = enc(omega)
        self.omega_inv = enc(pow(omega, -1, R_MOD))
        self.extended_omega = enc(ext_omega)
        self.extended_omega_inv = enc(pow(ext_omega, -1, R_MOD))
        self.g_coset = enc(zeta)
        self.g_coset_inv = enc(zeta * zeta % R_MOD)
        self.ifft_divisor = enc(pow(self.n, -1, R_MOD))
        self.extended_ifft_divisor = enc(pow(1 << ek, -1, R_MOD))
        orig, step = pow(zeta, self.n, R_MOD), pow(ext_omega, self.n, R_MOD)
        t, cur = [], orig
        while True:
            t.append(cur)
            cur = cur * step % R_MOD
            if cur == orig:
                break
        assert len(t) == 1 << (ek - k)
        self.t_evaluations = fr_encode([pow((v - 1) % R_MOD, -1, R_MOD) for v in t])

    def extended_len(self) -> int:
        return 1 << self.extended_k

    # ---- scalar helpers of `EvaluationDomain` the prover / verifier use next to the transforms (host big-int arithmetic, as the
    # reference computes them on the host) -------------------------------------------------------------------------------------------
    def rotate_omega(self, value: int, rotation: int) -> int:
        """`rotate_omega(value, Rotation(r))` = value * omega^r (r may be negative)"""
        w = self._omega if rotation >= 0 else pow(self._omega, -1, R_MOD)
        return value * pow(w, abs(rotation), R_MOD) % R_MOD

    def l_i_range(self, x: int, xn: int, rotations) -> list:
        """`l_i_range(x, x^n, rotations)`: the Lagrange basis polynomials l_i(X) = (X^n - 1) / (n (X omega^-i - 1)) evaluated at x for every
        rotation i of the range; a rotation whose omega^i equals x gives 1 (the reference's special case through batch inversion of zero)"""
        rotations = list(rotations)
        common = (xn - 1) * pow(self.n, -1, R_MOD) % R_MOD
        out = []
        for rot in rotations:
            wi = self.rotate_omega(1, rot)
            den = (x - wi) % R_MOD
            out.append(1 if den == 0 else common * wi % R_MOD * pow(den, -1, R_MOD) % R_MOD)
        return out

    @staticmethod
    def _p(a):
        return a.ctypes.data

    def lagrange_to_coeff(self, a: np.ndarray) -> np.ndarray:
        assert a.shape == (self.n, 4)
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        _lib.check(_lib.load().zkhip_ifft_scaled(self._p(a), self._p(self.omega_inv), self.k, self._p(self.ifft_divisor)))
        return a

    def coeff_to_extended(self, a: np.ndarray) -> np.ndarray:
        assert a.shape == (self.n, 4)
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.empty((self.extended_len(), 4), dtype=np.uint64)
        _lib.check(_lib.load().zkhip_coeff_to_extended(self._p(a), self.k, self._p(out), self.extended_k,
                                                       self._p(self.extended_omega), self._p(self.g_coset)))
        return out

    def extended_to_coeff(self, a: np.ndarray) -> np.ndarray:
        assert a.shape == (self.extended_len(), 4)
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        out_len = self.n * self.quotient_poly_degree
        out = np.empty((out_len, 4), dtype=np.uint64)
        _lib.check(_lib.load().zkhip_extended_to_coeff(self._p(a), self.extended_k, self._p(self.extended_omega_inv),
                                                       self._p(self.extended_ifft_divisor), self._p(self.g_coset),
                                                       self._p(out), out_len))
        return out

    def divide_by_vanishing_poly(self, a: np.ndarray) -> np.ndarray:
        assert a.shape == (self.extended_len(), 4)
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        _lib.check(_lib.load().zkhip_mul_periodic(self._p(a), a.shape[0], self._p(self.t_evaluations),
                                                  self.t_evaluations.shape[0]))
        return a
