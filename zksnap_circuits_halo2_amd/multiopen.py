"""`ProverGWC::create_proof` — the KZG multi-open argument of the reference's `gen_snark` path.

Mirror of [DEP] halo2-axiom `poly/kzg/multiopen/gwc.rs` + `gwc/prover.rs` as the reference reaches it:
`gen_snark::<.., ProverGWC<_>, ..>` (/root/reference/aggregator/src/wrapper.rs:59-60, 127-137; accumulation scheme `KzgAs<Bn256, Gwc19>`,
wrapper.rs:55).  For every distinct opening point z, in the order the points first appear among the queries:

    poly_batch = sum_i v^i p_i            eval_batch = sum_i v^i e_i             (the queries at z, in query order)
    W_z        = commit( kate_division(poly_batch - eval_batch, z) )

Everything runs on device-resident polynomials: the combination is one fused row program (`linear_combination_program`), the
quotient is `zkhip_fr_kate_division_device`, the commitment the prepared MSM.  The transcript is the host's business: the challenge v is
an argument and the witness commitments are returned (the reference writes each to the transcript as it is produced).
The bench path of the reference uses SHPLONK instead (halo2-base `gen_proof`); its building blocks are the same three kernels plus
`zkhip_fr_eval_polynomial_batch_device`, but its composition (rotation sets, interpolated evaluations) is not mirrored here.
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
            _lib.check(lib.zkhip_download(host.ctypes.data, out, host.nbytes))
        finally:
            lib.zkhip_free(out)
        for q, e in zip(qs, fr_decode(host)):
            q.eval = e


class ProverGWC:
    """`ProverGWC::new(params)` / `create_proof(transcript, queries)`; `commit` is a callable taking the device address of 2^k
    coefficients and returning the commitment (12 uint64 limbs, Jacobian) -- e.g. a prepared-table MSM over `params.g`"""

    def __init__(self, k: int, commit):
        self.k, self.n, self.commit = k, 1 << k, commit

    def create_proof(self, queries: Sequence[ProverQuery], v: int, stream: int = 0) -> List[np.ndarray]:
        lib = _lib.load()
        n = self.n
        evaluate_queries(queries, self.k, stream)
        batch, quot = C.c_void_p(), C.c_void_p()
        _lib.check(lib.zkhip_alloc(n * 32, C.byref(batch)))
        witnesses = []
        try:
            _lib.check(lib.zkhip_alloc(n * 32, C.byref(quot)))
            for z, qs in construct_intermediate_sets(queries):
                powers = [pow(v, i, R_MOD) for i in range(len(qs))]
                eval_batch = sum(p * q.eval for p, q in zip(powers, qs)) % R_MOD
                # poly_batch - eval_batch: the combination over all n rows, then the constant term alone (a 1-row program on the same buffer)
                E.linear_combination_program(powers).run_device([q.poly for q in qs], self.k, batch.value, stream=stream)
                fix = E.RowProgram()
                fix.emit(E.OP_SUB, 0, fix.column(0), fix.constant(eval_batch))
                fix.run_device([batch.value], 0, batch.value, stream=stream)
                zw = fr_encode([z])[0]
                _lib.check(lib.zkhip_fr_kate_division_device(batch, n, zw.ctypes.data, quot, stream))
                # the quotient has n - 1 coefficients; the commitment takes n scalars: clear the last one
                zero = np.zeros(4, dtype=np.uint64)
                _lib.check(lib.zkhip_sync())
                _lib.check(lib.zkhip_upload(C.c_void_p(quot.value + (n - 1) * 32), zero.ctypes.data, 32))
                witnesses.append(np.array(self.commit(quot.value), dtype=np.uint64).reshape(12))
        finally:
            _lib.check(lib.zkhip_sync())
            lib.zkhip_free(batch)
            if quot:
                lib.zkhip_free(quot)
        return witnesses
