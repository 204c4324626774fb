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
            W2 = prover.create_proof(queries, v)                     # a second proof reuses the prover's buffers
            assert all(np.array_equal(cref.jac_to_affine(a), cref.jac_to_affine(b)) for a, b in zip(W, W2))      # (Jacobian representatives differ run to run)
            # on a non-blocking side stream (every torch side stream is one): the host reads inside create_proof are ordered behind it
            import torch
            st = torch.cuda.Stream()
            fresh = [M.ProverQuery(points[pt], d_polys[pi].value) for pi, pt in plan]
            W3 = prover.create_proof(fresh, v, stream=st.cuda_stream)
            assert all(np.array_equal(cref.jac_to_affine(a), cref.jac_to_affine(b)) for a, b in zip(W, W3))
            assert [q.eval for q in fresh] == [q.eval for q in queries]
            prover.close()
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


def test_shplonk_create_proof_on_a_known_trapdoor_srs(lib, cref):
    """`ProverSHPLONK::create_proof` (the bench path's multi-open): rotation sets as the reference builds them, and the two commitments
    satisfy the SHPLONK opening equation in the exponent on an SRS whose trapdoor is known:
        Z_{T\\S_0}(u) (s - u) H' = sum_i v^i Z_{T\\S_i}(u) sum_j y^j (C_ij - R_ij(u) G) - Z_T(u) H
    together with h(s) Z-divisibility: for every set, sum_j y^j (P_ij - R_ij)(s) = Q_i(s) Z_i(s) is implied by H = [sum_i v^i Q_i(s)] G,
    which is checked directly against big-integer arithmetic."""
    k, s = 8, 0x0F1E2D3C4B5A6978
    n = 1 << k
    with Z.ParamsKZG.setup(k, s) as params:
        g = params.g.copy()
    polys = [cref.gen_scalars(3300 + i, n, 0) for i in range(5)]
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
        gen = O.SplitMix64(33)
        x = gen.fr()
        w = F.omega_for(k)
        px, pn, pp = x, x * w % R, x * pow(w, -1, R) % R
        # polys 0, 3: {x};  1: {x, wx};  2, 4: {x, wx, w^-1 x}: three rotation sets, in this order of first appearance
        plan = [(0, px), (1, px), (1, pn), (2, px), (2, pn), (2, pp), (3, px), (4, pp), (4, px), (4, pn)]
        queries = [M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan]
        y, v, u = gen.fr(), gen.fr(), gen.fr()
        sh = M.ProverSHPLONK(k, commit)
        H, Hp = sh.create_proof(queries, y, v, u)
        H2, Hp2 = sh.create_proof(queries, y, v, u)                  # buffers reused
        assert np.array_equal(cref.jac_to_affine(H), cref.jac_to_affine(H2)) and np.array_equal(cref.jac_to_affine(Hp), cref.jac_to_affine(Hp2))
        import torch
        st = torch.cuda.Stream()                                     # non-blocking: not ordered against the legacy default stream
        fresh = [M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan]
        H3, Hp3 = sh.create_proof(fresh, y, v, u, stream=st.cuda_stream)
        assert np.array_equal(cref.jac_to_affine(H), cref.jac_to_affine(H3)) and np.array_equal(cref.jac_to_affine(Hp), cref.jac_to_affine(Hp3))
        sh.close()
        sets, T = M.construct_rotation_sets(queries)
        ptr_to_idx = {d.value: i for i, d in enumerate(d_polys)}
        assert [[ptr_to_idx[p] for p in rs.polys] for rs in sets] == [[0, 3], [1], [2, 4]]
        assert [rs.points for rs in sets] == [[px], sorted([px, pn]), sorted([px, pn, pp])] and T == sorted([px, pn, pp])
        ints = [F.fr_decode(p) for p in polys]
        at_s = [O.eval_polynomial(c, s) for c in ints]
        Gp = cref.generator()
        # H = [h(s)] G with h(s) = sum_i v^i (sum_j y^j (P_ij(s) - R_ij(s))) / Z_i(s)
        hs = 0
        for i, rs in enumerate(sets):
            num = 0
            for j, p in enumerate(rs.polys):
                r_s = M._eval_small(M._interpolate(rs.points, rs.evals[j]), s)
                num = (num + pow(y, j, R) * (at_s[ptr_to_idx[p]] - r_s)) % R
            hs = (hs + pow(v, i, R) * num % R * pow(M._vanishing_at(rs.points, s), -1, R)) % R
        assert np.array_equal(cref.jac_to_affine(H), cref.jac_to_affine(cref.scalar_mul(hs, Gp)))
        # the opening equation, everything as multiples of G (the commitments C_ij = [P_ij(s)] G on this SRS)
        rhs = 0
        for i, rs in enumerate(sets):
            zi = M._vanishing_at([p for p in T if p not in rs.points], u)
            inner = 0
            for j, p in enumerate(rs.polys):
                r_u = M._eval_small(M._interpolate(rs.points, rs.evals[j]), u)
                inner = (inner + pow(y, j, R) * (at_s[ptr_to_idx[p]] - r_u)) % R
            rhs = (rhs + pow(v, i, R) * zi % R * inner) % R
        rhs = (rhs - M._vanishing_at(T, u) * hs) % R
        z0 = M._vanishing_at([p for p in T if p not in sets[0].points], u)
        lhs_point = cref.scalar_mul(z0 * (s - u) % R, cref.jac_to_affine(Hp))
        assert np.array_equal(cref.jac_to_affine(lhs_point), cref.jac_to_affine(cref.scalar_mul(rhs, Gp)))
        # an evaluation that does not belong to its polynomial is caught by the prover's own L(u) = 0 assertion
        bad = [M.ProverQuery(q.point, q.poly, q.eval) for q in queries]
        bad[4].eval = (bad[4].eval + 1) % R
        sh = M.ProverSHPLONK(k, commit)
        with pytest.raises(ArithmeticError):
            sh.create_proof(bad, y, v, u)
        sh.close()
    finally:
        _lib.check(lib.zkhip_release_bases(h))
        for ptr in d_polys + [d_g, d_out]:
            lib.zkhip_free(ptr)


def test_cpp_multiopen_mirror_matches_python(lib, cref, tmp_path):
    """The compiled-host mirror (include/zkhip.hpp: DeviceCommitter, gwc_create_proof, shplonk_create_proof) against the Python mirror on
    the same polynomials, queries and challenges: the same commitments (compared in affine form), and the L(u) = 0 assertion fires."""
    import os
    import struct
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "tests", "cpp", "multiopen_driver")
    assert os.path.exists(drv), "build it with __graft_entry__.build()"
    k, s = 10, 0x5A5A1234F00D
    n = 1 << k
    polys = [cref.gen_scalars(3500 + i, n, i % 2) for i in range(6)]
    gen = O.SplitMix64(35)
    x = gen.fr()
    w = F.omega_for(k)
    px, pn, pp, p2 = x, x * w % R, x * pow(w, -1, R) % R, x * w * w % R
    plan = [(0, px), (1, px), (1, pn), (2, pp), (2, px), (2, pn), (3, px), (4, pp), (4, px), (4, pn), (5, p2), (5, px)]
    y, v, u = gen.fr(), gen.fr(), gen.fr()
    fin, report = tmp_path / "in.bin", tmp_path / "report.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<IIIQ", k, len(polys), len(plan), s))
        for p in polys:
            f.write(p.tobytes())
        for pi, pt in plan:
            f.write(struct.pack("<I", pi))
            f.write(F.fr_encode([pt]).tobytes())
        f.write(F.fr_encode([y, v, u]).tobytes())
    subprocess.check_call([drv, str(fin), str(report)], timeout=300)
    rep = report.read_bytes()
    count, = struct.unpack("<I", rep[:4])
    pts = np.frombuffer(rep[4:4 + 64 * (count + 2)], dtype=np.uint64).reshape(count + 2, 8)
    flags, = struct.unpack("<Q", rep[4 + 64 * (count + 2):])
    assert flags == 1
    # the Python mirror on the same inputs
    with Z.ParamsKZG.setup(k, s) as params:
        g = params.g.copy()
    d_polys = []
    for p in polys:
        ptr = C.c_void_p()
        _lib.check(lib.zkhip_alloc(n * 32, C.byref(ptr)))
        _lib.check(lib.zkhip_upload(ptr, p.ctypes.data, n * 32))
        d_polys.append(ptr)
    d_g, d_out, h = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
    _lib.check(lib.zkhip_alloc(n * 64, C.byref(d_g)))
    _lib.check(lib.zkhip_upload(d_g, g.ctypes.data, n * 64))
    _lib.check(lib.zkhip_prepare_bases_device(d_g, n, C.byref(h)))
    _lib.check(lib.zkhip_alloc(96, C.byref(d_out)))

    def commit(d_coeffs):
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, C.c_void_p(d_coeffs), n, d_out, None))
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.zkhip_download(out.ctypes.data, d_out, 96))
        return out

    try:
        gwc = M.ProverGWC(k, commit)
        W = gwc.create_proof([M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan], v)
        gwc.close()
        sh = M.ProverSHPLONK(k, commit)
        H, Hp = sh.create_proof([M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan], y, v, u)
        sh.close()
        assert count == len(W) == 4
        for a, b in zip(list(W) + [H, Hp], pts):
            assert np.array_equal(cref.jac_to_affine(a), b)
    finally:
        _lib.check(lib.zkhip_release_bases(h))
        for ptr in d_polys + [d_g, d_out]:
            lib.zkhip_free(ptr)


def _c_queries(plan, d_polys, evals=None):
    qs = (_lib.ProverQueryC * len(plan))()
    for i, (pi, pt) in enumerate(plan):
        qs[i].point[:] = [int(x) for x in F.fr_encode([pt])[0]]
        qs[i].d_poly = d_polys[pi].value
        if evals is not None and evals[i] is not None:
            qs[i].eval[:] = [int(x) for x in F.fr_encode([evals[i]])[0]]
            qs[i].has_eval = 1
    return qs


def test_c_abi_multiopen_matches_the_python_provers(lib, cref):
    """zkhip_multiopen_gwc_device / zkhip_multiopen_shplonk_{begin, finish}_device -- the C++ provers behind `extern "C"`, on borrowed device
    addresses, committing against a REGISTERED base array as the Rust shim does -- against the Python provers on the same queries (which
    are checked against the oracle and the KZG equation above); with and without evaluations supplied by the caller; an evaluation that
    does not belong to its polynomial is refused by `finish`, which still releases the state"""
    k, s = 10, 0x5A5A1234F00D
    n = 1 << k
    polys = [cref.gen_scalars(3700 + i, n, i % 2) for i in range(6)]
    gen = O.SplitMix64(37)
    x = gen.fr()
    w = F.omega_for(k)
    px, pn, pp, p2 = x, x * w % R, x * pow(w, -1, R) % R, x * w * w % R
    plan = [(0, px), (1, px), (1, pn), (2, pp), (2, px), (2, pn), (3, px), (4, pp), (4, px), (4, pn), (5, p2), (5, px)]
    y, v, u = gen.fr(), gen.fr(), gen.fr()
    with Z.ParamsKZG.setup(k, s) as params:
        g = params.g.copy()
    d_polys = []
    for p in polys:
        ptr = C.c_void_p()
        _lib.check(lib.zkhip_alloc(n * 32, C.byref(ptr)))
        _lib.check(lib.zkhip_upload(ptr, p.ctypes.data, n * 32))
        d_polys.append(ptr)
    _lib.check(lib.zkhip_register_bases(g.ctypes.data, n))
    d_out = C.c_void_p()
    _lib.check(lib.zkhip_alloc(96, C.byref(d_out)))

    def commit(d_coeffs):
        _lib.check(lib.zkhip_msm_g1_registered_device(g.ctypes.data, C.c_void_p(d_coeffs), n, d_out, None))
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.zkhip_download(out.ctypes.data, d_out, 96))
        return out

    yw, vw, uw = (F.fr_encode([t])[0] for t in (y, v, u))          # kept alive: the calls below take their addresses
    try:
        gwc = M.ProverGWC(k, commit)
        W = gwc.create_proof([M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan], v)
        gwc.close()
        sh = M.ProverSHPLONK(k, commit)
        qpy = [M.ProverQuery(pt, d_polys[pi].value) for pi, pt in plan]
        H, Hp = sh.create_proof(qpy, y, v, u)
        sh.close()
        evals = [q.eval for q in qpy]                                   # filled in by the Python prover
        aff = lambda a: cref.jac_to_affine(np.ascontiguousarray(a))
        for supplied in (None, evals):
            qs = _c_queries(plan, d_polys, supplied)
            out = np.zeros((8, 12), dtype=np.uint64)
            cnt = C.c_size_t(0)
            _lib.check(lib.zkhip_multiopen_gwc_device(g.ctypes.data, k, qs, len(plan), vw.ctypes.data, out.ctypes.data, 8, C.byref(cnt)))
            assert cnt.value == len(W) == 4
            for a, b in zip(out[:4], W):
                assert np.array_equal(aff(a), aff(b))
            h_out, hp_out = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
            st = C.c_void_p()
            _lib.check(lib.zkhip_multiopen_shplonk_begin_device(g.ctypes.data, k, qs, len(plan), yw.ctypes.data, vw.ctypes.data, h_out.ctypes.data, C.byref(st)))
            assert st.value
            _lib.check(lib.zkhip_multiopen_shplonk_finish_device(st, uw.ctypes.data, hp_out.ctypes.data))
            assert np.array_equal(aff(h_out), aff(H)) and np.array_equal(aff(hp_out), aff(Hp))
        # room for fewer witnesses than there are points
        cnt = C.c_size_t(0)
        assert lib.zkhip_multiopen_gwc_device(g.ctypes.data, k, qs, len(plan), vw.ctypes.data, out.ctypes.data, 3, C.byref(cnt)) == -1 and cnt.value == 4
        # a wrong evaluation: begin succeeds (h is built from what it was given), finish refuses (L(u) != 0) and releases the state
        bad = list(evals)
        bad[1] = (bad[1] + 1) % R
        qs = _c_queries(plan, d_polys, bad)
        st = C.c_void_p()
        _lib.check(lib.zkhip_multiopen_shplonk_begin_device(g.ctypes.data, k, qs, len(plan), yw.ctypes.data, vw.ctypes.data, h_out.ctypes.data, C.byref(st)))
        assert lib.zkhip_multiopen_shplonk_finish_device(st, uw.ctypes.data, hp_out.ctypes.data) == -1
        assert b"L(u) != 0" in lib.zkhip_last_error()
        # abort; bad arguments
        st = C.c_void_p()
        _lib.check(lib.zkhip_multiopen_shplonk_begin_device(g.ctypes.data, k, _c_queries(plan, d_polys), len(plan), yw.ctypes.data, vw.ctypes.data,
                                                            h_out.ctypes.data, C.byref(st)))
        assert lib.zkhip_multiopen_shplonk_abort(st) == 0
        assert lib.zkhip_multiopen_shplonk_finish_device(None, uw.ctypes.data, hp_out.ctypes.data) == -1
        assert lib.zkhip_multiopen_shplonk_begin_device(g.ctypes.data, k, qs, 0, yw.ctypes.data, vw.ctypes.data, h_out.ctypes.data, C.byref(st)) == -1
        qs[3].d_poly = None
        assert lib.zkhip_multiopen_gwc_device(g.ctypes.data, k, qs, len(plan), vw.ctypes.data, out.ctypes.data, 8, C.byref(cnt)) == -1
    finally:
        _lib.check(lib.zkhip_unregister_bases(g.ctypes.data))
        for ptr in d_polys + [d_out]:
            lib.zkhip_free(ptr)
