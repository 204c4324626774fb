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

    def close(self) -> None:
        lib = _lib.load()
        lib.zkhip_unregister_bases(self.g.ctypes.data)
        if self.g_lagrange is not None:
            lib.zkhip_unregister_bases(self.g_lagrange.ctypes.data)
