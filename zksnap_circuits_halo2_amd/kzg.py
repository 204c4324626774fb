"""`ParamsKZG<Bn256>` mirror [DEP halo2-axiom poly/kzg/commitment.rs] (setup / read / write: section 8(f) row 4):
owns `g` / `g_lagrange`, pins them in HBM once (`zkhip_register_bases`) and commits with the GPU MSM."""
from __future__ import annotations

import numpy as np

from . import _lib
from .arithmetic import best_multiexp


def g_to_lagrange(g: np.ndarray, k: int) -> np.ndarray:
    """`g_to_lagrange(g_projective, k)` [DEP poly/kzg/commitment.rs]: the Lagrange-basis SRS of a monomial-basis SRS whose trapdoor is not
    known, g_lagrange[i] = (1/n) sum_j omega^(-i j) g[j] -- an inverse FFT over G1 points on the GPU (`zkhip_g_to_lagrange_device`)."""
    import ctypes as C

    n = 1 << k
    g = np.ascontiguousarray(g, dtype=np.uint64).reshape(n, 8)
    lib = _lib.load()
    d_in, d_out = C.c_void_p(), C.c_void_p()
    _lib.check(lib.zkhip_alloc(n * 64, C.byref(d_in)))
    try:
        _lib.check(lib.zkhip_alloc(n * 64, C.byref(d_out)))
        try:
            _lib.check(lib.zkhip_upload(d_in, g.ctypes.data, n * 64))
            _lib.check(lib.zkhip_g_to_lagrange_device(d_in, k, d_out, None))
            out = np.zeros((n, 8), dtype=np.uint64)
            _lib.check(lib.zkhip_download(out.ctypes.data, d_out, n * 64))
            return out
        finally:
            lib.zkhip_free(d_out)
    finally:
        lib.zkhip_free(d_in)


class ParamsKZG:
    def __init__(self, k: int, g: np.ndarray, g_lagrange: np.ndarray | None = None, g2: np.ndarray | None = None, s_g2: np.ndarray | None = None):
        self.k, self.n = k, 1 << k
        self.g = np.ascontiguousarray(g, dtype=np.uint64).reshape(self.n, 8)
        self.g_lagrange = None if g_lagrange is None else np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(self.n, 8)
        # G2Affine memory (16 u64), carried for the verifying key and the SRS file only (srs.py)
        self.g2 = None if g2 is None else np.ascontiguousarray(g2, dtype=np.uint64).reshape(16)
        self.s_g2 = None if s_g2 is None else np.ascontiguousarray(s_g2, dtype=np.uint64).reshape(16)
        lib = _lib.load()
        # The library recognises registered bases by host address (include/zkhip.h), so the registration must end before the arrays
        # are freed: `close()` / the context manager do it explicitly, and a finalizer does it when the object is collected without
        # either (the arrays are kept alive by the finalizer's own references until then).  The arrays are made read-only: the prepared
        # tables are built from their contents at registration.
        arrays = [self.g] + ([self.g_lagrange] if self.g_lagrange is not None else [])
        registered = []
        try:
            for a in arrays:
                _lib.check(lib.zkhip_register_bases(a.ctypes.data, self.n))
                registered.append(a)
        except Exception:
            for a in registered:
                lib.zkhip_unregister_bases(a.ctypes.data)
            raise
        # (np.ascontiguousarray hands back the caller's own array when it already is contiguous uint64: remember which arrays were
        # writeable and give that back when the registration ends, so that a caller's array is not left read-only for good)
        was_writeable = [bool(a.flags.writeable) for a in arrays]
        for a in arrays:
            a.flags.writeable = False
        import weakref

        self._finalizer = weakref.finalize(self, _unregister_arrays, arrays, was_writeable)

    @classmethod
    def setup(cls, k: int, s: int) -> "ParamsKZG":
        """`ParamsKZG::setup(k, rng)` [DEP poly/kzg/commitment.rs] with the trapdoor `s` given instead of drawn (what the reference's
        benches and `gen_srs` test parameters amount to: /root/reference/voter/benches/voter_circuit.rs:60): g[i] = [s^i] G and
        g_lagrange[i] = [(s^n - 1)/n * w^i / (s - w^i)] G, both computed on the GPU -- the scalar vectors with the Fr primitives
        (prefix product, row programs, batch inversion), the points with the fixed-base multiplication."""
        import ctypes as C

        from . import evaluation as E
        from .fields import R_MOD, fr_encode, omega_for

        n = 1 << k
        s %= R_MOD
        omega = omega_for(k)
        assert pow(s, n, R_MOD) != 1, "the trapdoor lies in the evaluation domain"
        lib = _lib.load()
        d_sc, d_tmp, d_pts = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for ptr, size in ((d_sc, n * 32), (d_tmp, n * 32), (d_pts, n * 64)):
            _lib.check(lib.zkhip_alloc(size, C.byref(ptr)))
        try:
            def points() -> np.ndarray:
                _lib.check(lib.zkhip_g1_fixed_base_mul_device(d_sc, n, d_pts, None))
                out = np.zeros((n, 8), dtype=np.uint64)
                _lib.check(lib.zkhip_download(out.ctypes.data, d_pts, n * 64))
                return out

            # s^i: exclusive prefix product of the constant vector (s, s, ...)
            const = np.ascontiguousarray(np.tile(fr_encode([s]), (n, 1)))
            _lib.check(lib.zkhip_upload(d_sc, const.ctypes.data, n * 32))
            _lib.check(lib.zkhip_fr_prefix_product_device(d_sc, n, d_sc, None))
            g = points()
            # Lagrange scalars: (s - w^i) -> inverse -> times w^i * (s^n - 1) / n
            den = E.RowProgram(omega=omega)
            den.emit(E.OP_SUB, 0, den.constant(s), E.RowProgram.ROWPOW)
            den.run_device([], k, d_tmp.value)
            _lib.check(lib.zkhip_fr_batch_invert_device(d_tmp, n, None))
            num = E.RowProgram(omega=omega)
            num.emit(E.OP_MUL, 0, E.RowProgram.ROWPOW, num.column(0))
            num.emit(E.OP_MUL, 0, E.RowProgram.reg(0), num.constant((pow(s, n, R_MOD) - 1) * pow(n, -1, R_MOD) % R_MOD))
            num.run_device([d_tmp.value], k, d_sc.value)
            g_lagrange = points()
        finally:
            for ptr in (d_sc, d_tmp, d_pts):
                lib.zkhip_free(ptr)
        from . import srs

        return cls(k, g, g_lagrange, srs.g2_encode(srs.G2_GENERATOR), srs.g2_encode(srs.g2_mul(s)))

    @classmethod
    def from_parts(cls, k: int, g: np.ndarray, g_lagrange: np.ndarray | None = None, g2: np.ndarray | None = None, s_g2: np.ndarray | None = None) -> "ParamsKZG":
        """`ParamsKZG::from_parts` [DEP]: the Lagrange basis is derived from g when it is not supplied"""
        return cls(k, g, g_to_lagrange(g, k) if g_lagrange is None else g_lagrange, g2, s_g2)

    def downsize(self, new_k: int) -> "ParamsKZG":
        """`ParamsKZG::downsize(k)` [DEP]: the first 2^k monomial-basis points, and their Lagrange basis re-derived with `g_to_lagrange`
        (the inverse FFT over G1 points on the GPU); returns a new object, as the mirror's arrays are pinned by address"""
        if new_k > self.k:
            raise ValueError("downsize: new_k > k")                                  # the reference asserts `new_k <= self.k`
        g = np.array(self.g[: 1 << new_k])
        return ParamsKZG(new_k, g, g_to_lagrange(g, new_k), self.g2, self.s_g2)

    def write(self, f) -> None:
        """`ParamsKZG::write` (SerdeFormat::RawBytes) [DEP]; layout in srs.py"""
        self.write_custom(f, "RawBytes")

    def write_custom(self, f, fmt: str, flag_layout: int = 0) -> None:
        """`ParamsKZG::write_custom(writer, format)` [DEP]: RawBytes / RawBytesUnchecked (the memory of the points) or Processed (every
        point compressed: the two tables on the GPU)"""
        from . import srs

        if self.g_lagrange is None or self.g2 is None or self.s_g2 is None:
            raise ValueError("write needs g_lagrange, g2 and s_g2")
        srs.write_params(f, self.k, self.g, self.g_lagrange, self.g2, self.s_g2, fmt, flag_layout)

    @classmethod
    def read(cls, f, check_points=None) -> "ParamsKZG":
        """`ParamsKZG::read` [DEP]: parses the file, verifies every point (one GPU pass per table: the RawBytes reader's check; pass a
        number to sample on the host instead, 0 for RawBytesUnchecked) and pins both tables in HBM"""
        from . import srs

        k, g, g_lagrange, g2, s_g2 = srs.read_params(f, check_points)
        return cls(k, g, g_lagrange, g2, s_g2)

    @classmethod
    def read_custom(cls, f, fmt: str, flag_layout: int = 0) -> "ParamsKZG":
        """`ParamsKZG::read_custom(reader, format)` [DEP]"""
        from . import srs

        k, g, g_lagrange, g2, s_g2 = srs.read_params(f, None, fmt=fmt, flag_layout=flag_layout)
        return cls(k, g, g_lagrange, g2, s_g2)

    def get_g2(self) -> np.ndarray:
        return self.g2

    def get_s_g2(self) -> np.ndarray:
        return self.s_g2

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

    def commit_device(self, d_poly: int, length: int, d_out: int, lagrange: bool = False, stream: int = 0) -> None:
        """`commit` / `commit_lagrange` of a polynomial that lives in HBM (`length` elements at device address d_poly), against the
        tables this object registered: the 96-byte Jacobian result is written to device address d_out, asynchronously on `stream`"""
        bases = self.g_lagrange if lagrange else self.g
        assert bases is not None and length <= self.n
        _lib.check(_lib.load().zkhip_msm_g1_registered_device(bases.ctypes.data, d_poly, length, d_out, stream))

    def commit_many_device(self, d_polys: int, length: int, count: int, stride: int, d_out: int, lagrange: bool = False, stream: int = 0) -> None:
        """`count` polynomials that live in HBM (polynomial i: `length` elements at d_polys + i * stride elements) committed by one call
        against the registered tables -- all advice columns of a circuit: small MSMs share launch sets (include/zkhip.h
        zkhip_msm_g1_registered_batch_device); `count` Jacobian results (96 bytes each) at device address d_out, asynchronously on `stream`"""
        bases = self.g_lagrange if lagrange else self.g
        assert bases is not None and length <= self.n and stride >= length
        _lib.check(_lib.load().zkhip_msm_g1_registered_batch_device(bases.ctypes.data, d_polys, length, count, stride, d_out, stream))

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
        """end the residency of g / g_lagrange (idempotent); also runs when the object is garbage-collected"""
        self._finalizer()

    def __enter__(self) -> "ParamsKZG":
        return self

    def __exit__(self, *exc) -> None:
        self.close()


def _unregister_arrays(arrays, was_writeable=()) -> None:
    try:
        lib = _lib.load()
    except Exception:   # noqa: BLE001  (interpreter shutdown)
        return
    for a in arrays:
        lib.zkhip_unregister_bases(a.ctypes.data)
    for a, w in zip(arrays, was_writeable):
        if w:
            try:
                a.flags.writeable = True
            except ValueError:   # a view of a read-only base
                pass
