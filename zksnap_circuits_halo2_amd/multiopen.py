"""The KZG multi-open arguments of the reference: `ProverGWC::create_proof` (the `gen_snark` path) and, below,
`ProverSHPLONK::create_proof` (the bench path).

Mirror of [DEP] halo2-axiom `poly/kzg/multiopen/gwc.rs` + `gwc/prover.rs` as the reference reaches it:
`gen_snark::<.., ProverGWC<_>, ..>` (/root/reference/aggregator/src/wrapper.rs:59-60, 127-137; accumulation scheme `KzgAs<Bn256, Gwc19>`,
wrapper.rs:55).  For every distinct opening point z, in the order the points first appear among the queries:

    poly_batch = sum_i v^i p_i            eval_batch = sum_i v^i e_i             (the queries at z, in query order)
    W_z        = commit( kate_division(poly_batch - eval_batch, z) )

Everything runs on device-resident polynomials: the combination is one fused row program (`linear_combination_program`), the
quotient is `zkhip_fr_kate_division_device`, the commitment the prepared MSM.  The transcript is the host's business: the challenge v is
an argument and the witness commitments are returned (the reference writes each to the transcript as it is produced).
The bench path of the reference uses SHPLONK instead (halo2-base `gen_proof`): `ProverSHPLONK` in the second half of this file.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib, evaluation as E
from .fields import R_MOD, fr_decode, fr_encode


@dataclass
class ProverQuery:
    """`ProverQuery { point, poly, eval }`: `poly` is the device address of 2^k coefficients (32 bytes each); `eval` may be None, in
    which case `evaluate_queries` fills it in (the prover computes these evaluations anyway and writes them to the transcript)"""
    point: int
    poly: int
    eval: int = None


def construct_intermediate_sets(queries: Sequence[ProverQuery]) -> List[Tuple[int, List[ProverQuery]]]:
    """`construct_intermediate_sets` of gwc.rs: queries grouped by point, points in order of first appearance, queries in query order"""
    sets: List[Tuple[int, List[ProverQuery]]] = []
    for q in queries:
        for point, qs in sets:
            if point == q.point % R_MOD:
                qs.append(q)
                break
        else:
            sets.append((q.point % R_MOD, [q]))
    return sets


def evaluate_queries(queries: Sequence[ProverQuery], k: int, stream: int = 0) -> None:
    """fills `eval` of every query that lacks it: one batched evaluation per distinct point (`eval_polynomial` on the device)"""
    lib = _lib.load()
    n = 1 << k
    for point, qs in construct_intermediate_sets([q for q in queries if q.eval is None]):
        out = C.c_void_p()
        _lib.check(lib.zkhip_alloc(len(qs) * 32, C.byref(out)))
        try:
            ptrs = (C.c_void_p * len(qs))(*[q.poly for q in qs])
            x = fr_encode([point])[0]
            _lib.check(lib.zkhip_fr_eval_polynomial_batch_device(ptrs, len(qs), n, x.ctypes.data, out, stream))
            host = np.zeros((len(qs), 4), dtype=np.uint64)
            _lib.check(lib.zkhip_stream_sync(stream))        # zkhip_download copies on the legacy default stream: not ordered behind a non-blocking `stream`
            _lib.check(lib.zkhip_download(host.ctypes.data, out, host.nbytes))
        finally:
            lib.zkhip_free(out)
        for q, e in zip(qs, fr_decode(host)):
            q.eval = e


def _sub_const_at(d_poly: int, index: int, value: int, stream: int = 0) -> None:
    """d_poly[index] -= value, enqueued on the stream (a one-row program: no host round trip)"""
    prog = E.RowProgram()
    prog.emit(E.OP_SUB, 0, prog.column(0), prog.constant(value))
    prog.run_device([d_poly + index * 32], 0, d_poly + index * 32, stream=stream)


def _zero_at(d_poly: int, index: int, stream: int = 0) -> None:
    prog = E.RowProgram()
    prog.emit(E.OP_MOV, 0, prog.constant(0))
    prog.run_device([], 0, d_poly + index * 32, stream=stream)


class _BufferPool:
    """n-coefficient device buffers kept between proofs (a prover object lives as long as its proving key)"""

    def __init__(self, n: int):
        self.n, self.free, self.all = n, [], []

    def take(self) -> int:
        if self.free:
            return self.free.pop()
        p = C.c_void_p()
        _lib.check(_lib.load().zkhip_alloc(self.n * 32, C.byref(p)))
        self.all.append(p)
        return p.value

    def give_back(self, taken: List[int]) -> None:
        self.free.extend(taken)

    def close(self) -> None:
        lib = _lib.load()
        for p in self.all:
            lib.zkhip_free(p)
        self.all, self.free = [], []


class ProverGWC:
    """`ProverGWC::new(params)` / `create_proof(transcript, queries)`; `commit` is a callable taking the device address of 2^k
    coefficients and returning the commitment (12 uint64 limbs, Jacobian) -- e.g. a prepared-table MSM over `params.g`.  `create_proof`
    enqueues on `stream` (any HIP stream, blocking or not) and drains it before every host read and before every call of `commit`."""

    def __init__(self, k: int, commit):
        self.k, self.n, self.commit = k, 1 << k, commit
        self.pool = _BufferPool(self.n)

    def close(self) -> None:
        self.pool.close()

    def create_proof(self, queries: Sequence[ProverQuery], v: int, stream: int = 0) -> List[np.ndarray]:
        lib = _lib.load()
        n = self.n
        evaluate_queries(queries, self.k, stream)
        batch, quot = self.pool.take(), self.pool.take()
        witnesses = []
        try:
            for z, qs in construct_intermediate_sets(queries):
                powers = [pow(v, i, R_MOD) for i in range(len(qs))]
                eval_batch = sum(p * q.eval for p, q in zip(powers, qs)) % R_MOD
                # poly_batch - eval_batch: the combination over all n rows, then the constant term alone (a one-row program on the same buffer)
                E.linear_combination_program(powers).run_device([q.poly for q in qs], self.k, batch, stream=stream)
                _sub_const_at(batch, 0, eval_batch, stream)
                zw = fr_encode([z])[0]
                _lib.check(lib.zkhip_fr_kate_division_device(C.c_void_p(batch), n, zw.ctypes.data, C.c_void_p(quot), stream))
                _zero_at(quot, n - 1, stream)           # the quotient has n - 1 coefficients; the commitment takes n scalars
                _lib.check(lib.zkhip_stream_sync(stream))     # `commit` runs on a stream of its own choosing: hand it finished coefficients
                witnesses.append(np.array(self.commit(quot), dtype=np.uint64).reshape(12))
        finally:
            _lib.check(lib.zkhip_sync())
            self.pool.give_back([batch, quot])
        return witnesses


# ---------------------------------------------------------------------------------------------------------------------------------
# SHPLONK: `ProverSHPLONK::create_proof` [DEP poly/kzg/multiopen/shplonk.rs + shplonk/prover.rs] -- the multi-open of the reference's
# bench path (halo2-base `gen_proof`: /root/reference/aggregator/benches/wrapper_circuit.rs:140, state_transition_circuit.rs:84,
# /root/reference/voter/benches/voter_circuit.rs:80).  Restated from the published algorithm (unpinned, as everything at this boundary):
#
#   rotation sets      polynomials grouped by the SET of points they are opened at (sets in order of first appearance of their first
#                      polynomial, polynomials in order of first appearance, the points of a set ascending as canonical integers: a BTreeSet)
#   R_ij               the interpolation of P_ij's evaluations over its set's points ("low degree equivalent")
#   h(X)  = sum_i v^i ( sum_j y^j (P_ij - R_ij) ) / Z_i(X),   Z_i = prod over the set's points (X - z)                    -> commit: H
#   L(X)  = sum_i v^i Z_{T \ S_i}(u) sum_j y^j (P_ij(X) - R_ij(u))  -  Z_T(u) h(X),     T = all points;   L(u) = 0
#   h'(X) = L(X) / (X - u) / Z_{T \ S_0}(u)                                                                              -> commit: H'
#
# Device work: the y- and v-combinations are fused row programs over the device-resident polynomials, the divisions
# `zkhip_fr_kate_division_device`, the two commitments prepared MSMs; the interpolations are host arithmetic on a handful of points.
# ---------------------------------------------------------------------------------------------------------------------------------
def _interpolate(points: Sequence[int], evals: Sequence[int]) -> List[int]:
    """`lagrange_interpolate`: coefficients (low to high) of the polynomial of degree < len(points) through (points[i], evals[i])"""
    m = len(points)
    coeffs = [0] * m
    for i in range(m):
        num = [1]
        den = 1
        for j in range(m):
            if j == i:
                continue
            num = [(a - points[j] * b) % R_MOD for a, b in zip([0] + num, num + [0])]      # num * (X - points[j])
            den = den * (points[i] - points[j]) % R_MOD
        scale = evals[i] * pow(den, -1, R_MOD) % R_MOD
        for t in range(m):
            coeffs[t] = (coeffs[t] + scale * num[t]) % R_MOD
    return coeffs


def _eval_small(coeffs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def _vanishing_at(points: Sequence[int], x: int) -> int:
    acc = 1
    for p in points:
        acc = acc * (x - p) % R_MOD
    return acc


@dataclass
class RotationSet:
    points: List[int]                      # ascending canonical integers
    polys: List[int]                       # device addresses, in order of first appearance
    evals: List[List[int]]                 # evals[j][t] = P_j(points[t])


def construct_rotation_sets(queries: Sequence[ProverQuery]) -> Tuple[List[RotationSet], List[int]]:
    """`construct_intermediate_sets` of shplonk.rs -> (rotation sets, super point set ascending)"""
    by_poly: dict = {}                                  # poly -> its points, in order of first appearance (dicts keep insertion order)
    evals: dict = {}
    for q in queries:
        by_poly.setdefault(q.poly, set()).add(q.point % R_MOD)
        evals.setdefault((q.poly, q.point % R_MOD), q.eval)          # the first query of a (poly, point) pair gives the evaluation
    sets_by_key: dict = {}                              # point set -> its polynomials, in order of first appearance
    for poly, pts in by_poly.items():
        sets_by_key.setdefault(frozenset(pts), []).append(poly)
    sets = list(sets_by_key.items())

    def get_eval(poly, point):
        return evals[(poly, point)]

    out = []
    for key, polys in sets:
        pts = sorted(key)
        out.append(RotationSet(pts, polys, [[get_eval(p, z) for z in pts] for p in polys]))
    return out, sorted({q.point % R_MOD for q in queries})


class ProverSHPLONK:
    def __init__(self, k: int, commit):
        self.k, self.n, self.commit = k, 1 << k, commit
        self.pool = _BufferPool(self.n)

    def close(self) -> None:
        self.pool.close()

    def _patch_low(self, d_poly: int, low: Sequence[int], stream: int = 0) -> None:
        """d_poly[t] -= low[t] for the first len(low) coefficients (the low degree equivalent has as many coefficients as the set has points)"""
        for t, val in enumerate(low):
            if val % R_MOD:
                _sub_const_at(d_poly, t, val, stream)

    def _divide(self, d_poly: int, d_tmp: int, roots: Sequence[int], stream: int) -> int:
        """`div_by_vanishing`: successive `kate_division`s, the quotient kept at n coefficients (zero-padded); returns the buffer that holds it"""
        lib = _lib.load()
        src, dst = d_poly, d_tmp
        for z in roots:
            zw = fr_encode([z])[0]
            _lib.check(lib.zkhip_fr_kate_division_device(C.c_void_p(src), self.n, zw.ctypes.data, C.c_void_p(dst), stream))
            _zero_at(dst, self.n - 1, stream)
            src, dst = dst, src
        return src

    def create_proof(self, queries: Sequence[ProverQuery], y: int, v: int, u: int, stream: int = 0):
        """-> (H, H') as 12-limb Jacobian commitments.  In the reference y and v are squeezed before h is committed and u after h has been
        written to the transcript; here all three are arguments."""
        lib = _lib.load()
        n = self.n
        evaluate_queries(queries, self.k, stream)
        sets, super_points = construct_rotation_sets(queries)
        bufs = []

        def alloc():
            bufs.append(self.pool.take())
            return bufs[-1]

        try:
            # ---- h(X): per set the y-combination of P - R, divided by the set's vanishing polynomial; then the v-combination -------------
            quotients = []
            for rs in sets:
                ypow = [pow(y, j, R_MOD) for j in range(len(rs.polys))]
                low = [_interpolate(rs.points, ev) for ev in rs.evals]                         # R_ij, len(points) coefficients each
                acc, tmp = alloc(), alloc()
                E.linear_combination_program(ypow).run_device(rs.polys, self.k, acc, stream=stream)
                self._patch_low(acc, [sum(yp * lo[t] for yp, lo in zip(ypow, low)) % R_MOD for t in range(len(rs.points))], stream)
                quotients.append(self._divide(acc, tmp, rs.points, stream))
            vpow = [pow(v, i, R_MOD) for i in range(len(sets))]
            h_x = alloc()
            E.linear_combination_program(vpow).run_device(quotients, self.k, h_x, stream=stream)
            _lib.check(lib.zkhip_stream_sync(stream))         # `commit` runs on a stream of its own choosing: hand it finished coefficients
            H = np.array(self.commit(h_x), dtype=np.uint64).reshape(12)
            # ---- L(X) and the final quotient ----------------------------------------------------------------------------------------------
            z_diffs = [_vanishing_at([p for p in super_points if p not in rs.points], u) for rs in sets]
            zt_eval = _vanishing_at(super_points, u)
            norm = pow(z_diffs[0], -1, R_MOD)                                                  # "normalize by the coefficient of the first polynomial"
            cols, coeffs, const = [], [], 0
            for i, rs in enumerate(sets):
                for j, poly in enumerate(rs.polys):
                    c = vpow[i] * z_diffs[i] % R_MOD * pow(y, j, R_MOD) % R_MOD * norm % R_MOD
                    cols.append(poly)
                    coeffs.append(c)
                    r_eval = _eval_small(_interpolate(rs.points, rs.evals[j]), u)               # R_ij(u)
                    const = (const + c * r_eval) % R_MOD
            cols.append(h_x)
            coeffs.append((-zt_eval * norm) % R_MOD)
            l_x, tmp = alloc(), alloc()
            E.linear_combination_program(coeffs).run_device(cols, self.k, l_x, stream=stream)
            self._patch_low(l_x, [const], stream)
            # the reference's debug assertion: L(u) = 0 (the result lands in the top coefficient slot of the scratch buffer, which the
            # division below overwrites)
            uw = fr_encode([u])[0]
            chk = tmp + (n - 1) * 32
            _lib.check(lib.zkhip_fr_eval_polynomial_device(C.c_void_p(l_x), n, uw.ctypes.data, C.c_void_p(chk), stream))
            res = np.zeros(4, dtype=np.uint64)
            _lib.check(lib.zkhip_stream_sync(stream))
            _lib.check(lib.zkhip_download(res.ctypes.data, C.c_void_p(chk), 32))
            if res.any():
                raise ArithmeticError("SHPLONK: L(u) != 0 -- inconsistent queries (an evaluation does not match its polynomial)")
            final = self._divide(l_x, tmp, [u], stream)
            _lib.check(lib.zkhip_stream_sync(stream))
            Hp = np.array(self.commit(final), dtype=np.uint64).reshape(12)
            return H, Hp
        finally:
            _lib.check(lib.zkhip_sync())
            self.pool.give_back(bufs)
