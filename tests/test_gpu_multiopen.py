"""GPU (-m gpu): `ProverGWC::create_proof` (the multi-open argument of the reference's gen_snark path, /root/reference/aggregator/src/wrapper.rs:59-60,
127-137) composed from the device entry points (multiopen.py), checked two ways: the witness commitments equal the oracle's
(kate_division + best_multiexp of the same combinations), and -- on an SRS whose trapdoor s is known -- every opening satisfies the KZG
equation in the exponent, (s - z) W = sum_i v^i C_i - [sum_i v^i e_i] G, which is what the pairing check of the verifier tests."""
import ctypes as C

import numpy as np
import pytest

import zksnap_circuits_halo2_amd as Z
from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F, multiopen as M

pytestmark = pytest.mark.gpu
R = O.R_MOD


def test_gwc_create_proof_on_a_known_trapdoor_srs(lib, cref):
    k, s = 9, 0x1F2E3D4C5B6A7988
    n = 1 << k
    with Z.ParamsKZG.setup(k, s) as params:
        g = params.g.copy()
        polys = [cref.gen_scalars(3100 + i, n, i % 2) for i in range(6)]
        d_polys = []
        for p in polys:
            ptr = C.c_void_p()
            _lib.check(lib.zkhip_alloc(n * 32, C.byref(ptr)))
            _lib.check(lib.zkhip_upload(ptr, p.ctypes.data, n * 32))
            d_polys.append(ptr)
        d_g = C.c_void_p()
        _lib.check(lib.zkhip_alloc(n * 64, C.byref(d_g)))
        _lib.check(lib.zkhip_upload(d_g, g.ctypes.data, n * 64))
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device(d_g, n, C.byref(h)))
        d_out = C.c_void_p()
        _lib.check(lib.zkhip_alloc(96, C.byref(d_out)))

        def commit(d_coeffs):
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, C.c_void_p(d_coeffs), n, d_out, None))
            out = np.zeros(12, dtype=np.uint64)
            _lib.check(lib.zkhip_download(out.ctypes.data, d_out, 96))
            return out

        try:
            gen = O.SplitMix64(31)
            x = gen.fr()
            w = F.omega_for(k)
            points = [x, x * w % R, x * pow(w, -1, R) % R]                       # x, omega x, omega^-1 x: the rotations a halo2 circuit opens at
            # query order mixes the points, as `ProverQuery`s of advice / fixed / permutation polynomials do
            plan = [(0, 0), (1, 0), (2, 1), (0, 1), (3, 0), (4, 2), (5, 0), (1, 2)]
            queries = [M.ProverQuery(points[pt], d_polys[pi].value) for pi, pt in plan]
            v = gen.fr()
            prover = M.ProverGWC(k, commit)
            W = prover.create_proof(queries, v)
            sets = M.construct_intermediate_sets(queries)
            assert [z for z, _ in sets] == points and [len(qs) for _, qs in sets] == [4, 2, 2] and len(W) == 3
            # evaluations filled in by the prover = the oracle's eval_polynomial
            for (pi, pt), q in zip(plan, queries):
                assert q.eval == F.fr_decode(cref.eval_polynomial(polys[pi], F.fr_encode([points[pt]])[0]))[0]
            Gp = cref.generator()
            for (z, qs), wj in zip(sets, W):
                idx = [next(pi for (pi, pt), qq in zip(plan, queries) if qq is q) for q in qs]
                pw = [pow(v, i, R) for i in range(len(qs))]
                # 1. against the oracle: the same combination, division and commitment
                comb = [0] * n
                for i, pi in zip(pw, idx):
                    for j, c in enumerate(F.fr_decode(polys[pi])):
                        comb[j] = (comb[j] + i * c) % R
                e_batch = sum(i * q.eval for i, q in zip(pw, qs)) % R
                comb[0] = (comb[0] - e_batch) % R
                quot = cref.kate_division(F.fr_encode(comb), F.fr_encode([z])[0])
                exp = cref.jac_to_affine(cref.best_multiexp(np.ascontiguousarray(quot), np.ascontiguousarray(g[: n - 1]), 2))
                assert np.array_equal(cref.jac_to_affine(wj), exp)
                # 2. the KZG opening equation in the exponent (known trapdoor): (s - z) W = sum v^i C_i - [sum v^i e_i] G
                lhs = cref.scalar_mul((s - z) % R, cref.jac_to_affine(wj))
                ps = 0
                for i, pi in zip(pw, idx):
                    ps = (ps + i * O.eval_polynomial(F.fr_decode(polys[pi]), s)) % R
                rhs = cref.scalar_mul((ps - e_batch) % R, Gp)
                assert np.array_equal(cref.jac_to_affine(lhs), cref.jac_to_affine(rhs))
        finally:
            _lib.check(lib.zkhip_release_bases(h))
            for ptr in d_polys + [d_g, d_out]:
                lib.zkhip_free(ptr)
