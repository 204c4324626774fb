"""`ParamsKZG<Bn256>` commit surface mirror [DEP halo2-axiom poly/kzg/commitment.rs]:
owns `g` / `g_lagrange`, pins them in HBM once (`zkhip_register_bases`) and commits with the GPU MSM."""
from __future__ import annotations

import numpy as np

from . import _lib
from .arithmetic import best_multiexp


class ParamsKZG:
    def __init__(self, k: int, g: np.ndarray, g_lagrange: np.ndarray | None = None):
        self.k, self.n = k, 1 << k
        self.g = np.ascontiguousarray(g, dtype=np.uint64).reshape(self.n, 8)
        self.g_lagrange = None if g_lagrange is None else np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(self.n, 8)
        lib = _lib.load()
        _lib.check(lib.zkhip_register_bases(self.g.ctypes.data, self.n))
        if self.g_lagrange is not None:
            _lib.check(lib.zkhip_register_bases(self.g_lagrange.ctypes.data, self.n))

    def get_g(self) -> np.ndarray:
        return self.g

    def commit(self, poly: np.ndarray) -> np.ndarray:
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        assert poly.shape[0] <= self.n
        return best_multiexp(poly, self.g[: poly.shape[0]])

    def commit_lagrange(self, poly: np.ndarray) -> np.ndarray:
        assert self.g_lagrange is not None
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        assert poly.shape[0] <= self.n
        return best_multiexp(poly, self.g_lagrange[: poly.shape[0]])

    def commit_many(self, polys: np.ndarray, lagrange: bool = False) -> np.ndarray:
        """Commit to K polynomials of equal length at once ((K, len, 4) uint64) -> (K, 12) uint64: one launch set."""
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        assert polys.ndim == 3 and polys.shape[2] == 4 and polys.shape[1] <= self.n
        bases = self.g_lagrange if lagrange else self.g
        assert bases is not None
        out = np.zeros((polys.shape[0], 12), dtype=np.uint64)
        _lib.check(_lib.load().zkhip_msm_g1_batch(polys.ctypes.data, bases.ctypes.data, polys.shape[1], polys.shape[0], out.ctypes.data))
        return out

    def close(self) -> None:
        lib = _lib.load()
        lib.zkhip_unregister_bases(self.g.ctypes.data)
        if self.g_lagrange is not None:
            lib.zkhip_unregister_bases(self.g_lagrange.ctypes.data)
