"""GPU (-m gpu): parity of the HIP path (through the C ABI) with the oracle.  Bit-exact: field / NTT outputs are
compared limb for limb, MSM outputs after affine normalisation (Jacobian representatives are not unique)."""
import json
import os
import random

import numpy as np
import pytest

import zksnap_circuits_halo2_amd as Z
from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
unhex = lambda s: [int(s[i:i + 16], 16) for i in range(0, len(s), 16)]


def aff(cref, xyz):
    """canonical affine limbs of a Jacobian result (also checks the limbs are canonical Montgomery values)."""
    return cref.jac_to_affine(np.ascontiguousarray(xyz))


def structured_expect(cref, sc, t0, d):
    return cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, t0, d), cref.generator()))


# ---------------------------------------------------------------- rows a1 / a2: field and curve layer
@pytest.mark.parametrize("field", [0, 1])
def test_field_ops_vs_oracle(lib, cref, field):
    p = O.Q_MOD if field == 0 else O.R_MOD
    n = 1 << 16
    g = O.SplitMix64(17 + field)
    a = np.array([O.limbs4(g.fr() % p) for _ in range(n)], dtype=np.uint64)   # arbitrary canonical limb patterns
    b = np.array([O.limbs4(g.fr() % p) for _ in range(n)], dtype=np.uint64)
    edge = [0, 1, p - 1, p - 2, (p - 1) // 2, O.to_mont(1, p), O.to_mont(p - 1, p), (1 << 253) % p]
    for i, e in enumerate(edge):
        a[i] = O.limbs4(e)
        b[i] = O.limbs4(edge[(3 * i + 1) % len(edge)])
    for op in range(4):
        out = np.zeros_like(a)
        _lib.check(lib.zkhip_test_field_op(field, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n))
        assert np.array_equal(out, cref.field_op(field, op, a, b)), f"field {field} op {op}"


def test_curve_ops_vs_oracle(lib, cref):
    n = 2048
    A, _, _ = cref.gen_bases(5, n)
    B, _, _ = cref.gen_bases(6, n)
    A[3] = 0; B[4] = 0; A[5] = 0; B[5] = 0          # identities on either / both sides
    B[6] = A[6]                                      # doubling through the mixed add
    B[7] = A[7]; B[7, 4:8] = F.g1_encode([O.neg(O.affine_from_limbs([int(x) for x in A[7]]))])[0, 4:8]   # P + (-P)
    pts = lambda M: [O.affine_from_limbs([int(x) for x in r]) for r in M[:64]]
    for op, f in ((0, lambda P, Q: O.add(P, Q)), (1, lambda P, Q: O.add(P, P)), (2, lambda P, Q: O.add(P, O.neg(Q)))):
        out = np.zeros((n, 12), dtype=np.uint64)
        _lib.check(lib.zkhip_test_g1_op(op, A.ctypes.data, B.ctypes.data, out.ctypes.data, n))
        got = [F.g1_decode_jacobian(out[i]) for i in range(64)]
        assert got == [f(P, Q) for P, Q in zip(pts(A), pts(B))], f"g1 op {op}"
        # the rest against the C oracle: G1Affine -> G1 (z = 1), then jac_add / doubling
        one = np.array(O.limbs4(O.to_mont(1, O.Q_MOD)), dtype=np.uint64)
        tojac = lambda r: np.concatenate([r, one]) if r.any() else np.zeros(12, dtype=np.uint64)
        for i in range(64, n, 7):
            ja, jb = tojac(A[i]), tojac(B[i])
            if op == 1:
                jb = ja
            elif op == 2 and jb.any():
                jb = jb.copy()
                jb[4:8] = cref.field_op(0, 2, np.zeros((1, 4), dtype=np.uint64), jb[4:8].reshape(1, 4))[0]
            assert np.array_equal(cref.jac_to_affine(out[i]), cref.jac_to_affine(cref.jac_add(ja, jb))), (op, i)
        assert F.g1_decode_jacobian(out[5]) is None or op == 1


def test_quad_cooperative_curve_ops_vs_oracle(lib, cref):
    """the 4-lane formulas of the reduction tail (csrc/ec_quad.hpp): 2a + 2b (general add of two non-affine points) and 4a,
    incl. identities on either side, equal points (the doubling branch) and opposite points"""
    n = 1501                                          # ragged: the last workgroup is partly empty
    A, _, _ = cref.gen_bases(15, n)
    B, _, _ = cref.gen_bases(16, n)
    A[3] = 0; B[4] = 0; A[5] = 0; B[5] = 0
    B[6] = A[6]
    B[7] = A[7]; B[7, 4:8] = F.g1_encode([O.neg(O.affine_from_limbs([int(x) for x in A[7]]))])[0, 4:8]
    one = np.array(O.limbs4(O.to_mont(1, O.Q_MOD)), dtype=np.uint64)
    tojac = lambda r: np.concatenate([r, one]) if r.any() else np.zeros(12, dtype=np.uint64)
    dbl = lambda j: cref.jac_add(j, j)
    for op in (3, 4):
        out = np.zeros((n, 12), dtype=np.uint64)
        _lib.check(lib.zkhip_test_g1_op(op, A.ctypes.data, B.ctypes.data, out.ctypes.data, n))
        for i in list(range(0, 40)) + list(range(40, n, 13)) + [n - 1]:
            ja, jb = tojac(A[i]), tojac(B[i])
            exp = cref.jac_add(dbl(ja), dbl(jb)) if op == 3 else dbl(dbl(ja))
            assert np.array_equal(cref.jac_to_affine(out[i]), cref.jac_to_affine(exp)), (op, i)
        pts = [O.affine_from_limbs([int(x) for x in r]) for r in A[:8]], [O.affine_from_limbs([int(x) for x in r]) for r in B[:8]]
        for i in range(8):
            P, Q = pts[0][i], pts[1][i]
            exp = O.add(O.add(P, P), O.add(Q, Q)) if op == 3 else O.add(O.add(P, P), O.add(P, P))
            assert F.g1_decode_jacobian(out[i]) == exp, (op, i)


# ---------------------------------------------------------------- row a3: best_multiexp
def test_msm_golden_vectors(cref):
    data = json.load(open(os.path.join(GOLD, "msm_g1.json")))
    for case in data["msm"]:
        sc = np.array([unhex(s) for s in case["scalars"]], dtype=np.uint64).reshape(-1, 4)
        bs = np.array([unhex(s) for s in case["bases"]], dtype=np.uint64).reshape(-1, 8)
        got = aff(cref, Z.best_multiexp(sc, bs))
        assert [int(x) for x in got] == unhex(case["expected_affine"]), case["name"]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 31, 32, 33, 100, 257, 1000, 4097, 1 << 13, (1 << 15) + 5, 1 << 16])
@pytest.mark.parametrize("kind", [0, 1])
def test_msm_vs_reference_algorithm(cref, n, kind):
    """same seeded inputs through the HIP path and through the C restatement of best_multiexp (8 threads)."""
    bases, t0, d = cref.gen_bases(100 + n, n)
    sc = cref.gen_scalars(200 + n + kind, n, kind)
    if n >= 100:
        sc[3] = 0; sc[4] = F.fr_encode([1])[0]; sc[5] = F.fr_encode([O.R_MOD - 1])[0]; sc[6] = F.fr_encode([1 << 253])[0]
        bases[9] = 0                                   # identity base with a non-zero scalar
        bases[11] = bases[10]; sc[11] = sc[10]         # repeated (base, scalar): forces a doubling inside a bucket
    got = aff(cref, Z.best_multiexp(sc, bases))
    assert np.array_equal(got, aff(cref, cref.best_multiexp(sc, bases, 8)))
    if n < 100:
        assert np.array_equal(got, structured_expect(cref, sc, t0, d))


def test_msm_rejects_length_mismatch():
    with pytest.raises(AssertionError):
        Z.best_multiexp(np.zeros((3, 4), dtype=np.uint64), np.zeros((4, 8), dtype=np.uint64))


@pytest.mark.parametrize("c", list(range(2, 17)))
def test_msm_every_window_size(lib, cref, c):
    import torch

    n = 3000
    bases, t0, d = cref.gen_bases(7, n)
    sc = cref.gen_scalars(8 + c, n, c % 2)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    dbs = torch.from_numpy(bases.view(np.int64)).cuda()
    dout = torch.zeros(12, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_msm_g1_device_c(dsc.data_ptr(), dbs.data_ptr(), n, dout.data_ptr(), c, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = aff(cref, dout.cpu().numpy().view(np.uint64))
    assert np.array_equal(got, structured_expect(cref, sc, t0, d))


@pytest.mark.parametrize("c", [0, 2, 4, 8, 13, 16])
def test_msm_general_path_glv_edge_scalars(lib, cref, c):
    """the general path splits k = k1 + lambda k2 (csrc/msm.hip k_digits_glv).  Scalars chosen by their HALVES: magnitudes that put a window at
    exactly 2^(c-1) (the digit -2^(c-1) of a negative half must not wrap in int16), halves of either sign, zero halves, the largest magnitudes
    the decomposition can produce, k = 0 / 1 / r - 1, and lambda itself (k1 = 0, k2 = 1)"""
    import torch

    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    r = O.R_MOD
    halves = [0, 1, -1, 1 << 15, -(1 << 15), (1 << 15) + (1 << 31), -((1 << 15) + (1 << 31) + (1 << 47)), (1 << 7), -(1 << 7), (1 << 3) | (1 << 7) | (1 << 11),
              (1 << 126) + 12345, -((1 << 126) + 999), 0x7fff7fff7fff7fff7fff7fff7fff7fff, -0x7fff7fff7fff7fff7fff7fff7fff7fff, 0x80008000800080008000800080008000 >> 2,
              -(0x80008000800080008000800080008000 >> 2)]
    ks = [0, 1, r - 1, lam, r - lam, (lam * lam) % r]
    ks += [(a + lam * b) % r for a in halves for b in halves]
    g = O.SplitMix64(4242 + c)
    ks += [g.fr() for _ in range(64)]
    n = len(ks)
    bases, t0, d = cref.gen_bases(21, n)
    sc = F.fr_encode(ks)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    dbs = torch.from_numpy(bases.view(np.int64)).cuda()
    dout = torch.zeros(12, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_msm_g1_device_c(dsc.data_ptr(), dbs.data_ptr(), n, dout.data_ptr(), c, None))
    torch.cuda.synchronize()
    assert np.array_equal(aff(cref, dout.cpu().numpy().view(np.uint64)), structured_expect(cref, sc, t0, d))
    # one scalar at a time: a wrong digit cannot hide behind a cancellation
    for i in (6, 7, 10, 11, 12, 40, 41, 100, 101, 180, 250):
        if i >= n:
            continue
        _lib.check(lib.zkhip_msm_g1_device_c(dsc.data_ptr() + 32 * i, dbs.data_ptr() + 64 * i, 1, dout.data_ptr(), c, None))
        torch.cuda.synchronize()
        got = aff(cref, dout.cpu().numpy().view(np.uint64))
        one = np.ascontiguousarray(sc[i:i + 1])
        assert np.array_equal(got, aff(cref, cref.best_multiexp(one, np.ascontiguousarray(bases[i:i + 1]), 1))), (c, i, hex(ks[i]))


def test_msm_heavy_buckets_and_degenerate_inputs(cref):
    n = 1 << 15
    bases, t0, d = cref.gen_bases(31, 64)
    bases = np.ascontiguousarray(np.tile(bases, (n // 64, 1)))      # every base repeated 512 times
    one = F.fr_encode([1])[0]
    for name, sc in (("all ones", np.tile(one, (n, 1))), ("all equal", np.tile(cref.gen_scalars(1, 1, 0), (n, 1))),
                     ("witness-like", cref.gen_scalars(9, n, 1)), ("all zero", np.zeros((n, 4), dtype=np.uint64))):
        sc = np.ascontiguousarray(sc, dtype=np.uint64)
        got = aff(cref, Z.best_multiexp(sc, bases))
        assert np.array_equal(got, aff(cref, cref.best_multiexp(sc, bases, 8))), name
    # all bases identity
    got = Z.best_multiexp(cref.gen_scalars(2, 500, 0), np.zeros((500, 8), dtype=np.uint64))
    assert F.g1_decode_jacobian(got) is None


def test_msm_registered_bases_and_subranges(lib, cref):
    n = 5000
    bases, t0, d = cref.gen_bases(77, n)
    params = Z.ParamsKZG(13, np.ascontiguousarray(np.vstack([bases, np.zeros((8192 - n, 8), dtype=np.uint64)])))
    try:
        sc = cref.gen_scalars(78, n, 0)
        full = aff(cref, params.commit(sc))
        assert np.array_equal(full, structured_expect(cref, sc, t0, d))
        # prefix of the registered array (commit of a shorter polynomial) and an interior slice
        assert np.array_equal(aff(cref, params.commit(sc[:1234])), aff(cref, cref.best_multiexp(sc[:1234], bases[:1234], 4)))
        sl = params.get_g()[100:900]
        assert np.array_equal(aff(cref, Z.best_multiexp(sc[:800], sl)), aff(cref, cref.best_multiexp(sc[:800], np.ascontiguousarray(bases[100:900]), 4)))
    finally:
        params.close()


@pytest.mark.parametrize("n", [1, 7, 300, 5000, 70001])
def test_msm_prepared_device_path(lib, cref, n):
    """fixed-base table path (zkhip_prepare_bases_device): full range, sub-ranges, identity entries, both scalar kinds."""
    import ctypes as C

    import torch

    bases, t0, d = cref.gen_bases(900 + n, n)
    if n > 100:
        bases[17] = 0
    dbs = torch.from_numpy(bases.view(np.int64)).cuda()
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(dbs.data_ptr(), n, C.byref(h)))
    try:
        for kind, (off, m) in enumerate([(0, n), (n // 3, n - n // 3), (0, max(1, n // 2))]):
            sc = cref.gen_scalars(950 + n + kind, m, kind % 2)
            dsc = torch.from_numpy(sc.view(np.int64)).cuda()
            dout = torch.zeros(12, dtype=torch.int64, device="cuda")
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, off, dsc.data_ptr(), m, dout.data_ptr(), None))
            torch.cuda.synchronize()
            got = aff(cref, dout.cpu().numpy().view(np.uint64))
            assert np.array_equal(got, aff(cref, cref.best_multiexp(sc, np.ascontiguousarray(bases[off:off + m]), 8))), (n, off, m)
        if n > 1000:   # one bucket holding every entry: exercises the combine tree (n/64 partials in a single bucket)
            sc = np.ascontiguousarray(np.tile(F.fr_encode([1])[0], (n, 1)))
            dsc = torch.from_numpy(sc.view(np.int64)).cuda()
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, dout.data_ptr(), None))
            torch.cuda.synchronize()
            assert np.array_equal(aff(cref, dout.cpu().numpy().view(np.uint64)), aff(cref, cref.best_multiexp(sc, bases, 8)))
        assert lib.zkhip_msm_g1_prepared_device(h, 1, dsc.data_ptr(), n, dout.data_ptr(), None) == -1   # range check
    finally:
        _lib.check(lib.zkhip_release_bases(h))


@pytest.mark.parametrize("c", [3, 9, 13, 16, 17, 18, 19, 20])
def test_msm_prepared_every_window_including_wide(lib, cref, c):
    """prepared path with an explicit window size; c > 16 takes the two-level (coarse groups + fine LDS histogram) sort."""
    import ctypes as C

    import torch

    n = 20011
    bases, t0, d = cref.gen_bases(1200 + c, n)
    bases[5] = 0
    dbs = torch.from_numpy(bases.view(np.int64)).cuda()
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device_c(dbs.data_ptr(), n, c, C.byref(h)))
    try:
        one = F.fr_encode([1])[0]
        cases = [cref.gen_scalars(1300 + c, n, 0), cref.gen_scalars(1400 + c, n, 1), np.ascontiguousarray(np.tile(one, (n, 1))),
                 np.ascontiguousarray(np.tile(F.fr_encode([O.R_MOD - 1])[0], (n, 1)))]
        for sc in cases:
            dsc = torch.from_numpy(sc.view(np.int64)).cuda()
            dout = torch.zeros(12, dtype=torch.int64, device="cuda")
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, dout.data_ptr(), None))
            torch.cuda.synchronize()
            assert np.array_equal(aff(cref, dout.cpu().numpy().view(np.uint64)), aff(cref, cref.best_multiexp(sc, bases, 8)))
    finally:
        _lib.check(lib.zkhip_release_bases(h))
    assert lib.zkhip_prepare_bases_device_c(dbs.data_ptr(), n, 21, C.byref(h)) == -1


@pytest.mark.parametrize("n,batch,pad", [(1, 3, 0), (100, 5, 7), (4096, 9, 0), (8192, 40, 16), (70001, 3, 1)])
def test_msm_prepared_batch_vs_reference_algorithm(lib, cref, n, batch, pad):
    """K scalar vectors against one prepared SRS in one launch set; every result against the reference algorithm."""
    import ctypes as C

    import torch

    bases, t0, d = cref.gen_bases(1500 + n, n)
    if n > 50:
        bases[7] = 0
    dbs = torch.from_numpy(bases.view(np.int64)).cuda()
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(dbs.data_ptr(), n, C.byref(h)))
    try:
        stride = n + pad
        host = np.zeros((batch * stride, 4), dtype=np.uint64)
        vecs = []
        for k in range(batch):
            sc = cref.gen_scalars(1600 + 17 * n + k, n, k % 2)
            if k == 1:
                sc[:] = 0                                   # an all-zero column
            if k == 2 and n > 1:
                sc[:] = F.fr_encode([1])[0]                 # an all-ones column (one heavy bucket)
            vecs.append(sc)
            host[k * stride:k * stride + n] = sc
        dsc = torch.from_numpy(host.view(np.int64)).cuda()
        dout = torch.zeros(batch * 12, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, dsc.data_ptr(), n, batch, stride, dout.data_ptr(), None))
        torch.cuda.synchronize()
        got = dout.cpu().numpy().view(np.uint64).reshape(batch, 12)
        for k in range(batch):
            assert np.array_equal(aff(cref, got[k]), aff(cref, cref.best_multiexp(vecs[k], bases, 8))), k
    finally:
        _lib.check(lib.zkhip_release_bases(h))


def test_commit_many_and_host_batch_ntt(lib, cref):
    """host-buffer batch entry points: ParamsKZG.commit_many (registered SRS) and zkhip_ntt_fr_batch."""
    k, n, K = 11, 1 << 11, 6
    bases, t0, d = cref.gen_bases(1700, n)
    params = Z.ParamsKZG(k, bases)
    try:
        polys = np.stack([cref.gen_scalars(1701 + i, n, i % 2) for i in range(K)])
        got = params.commit_many(polys)
        for i in range(K):
            assert np.array_equal(aff(cref, got[i]), structured_expect(cref, polys[i], t0, d)), i
        short = params.commit_many(np.ascontiguousarray(polys[:, :1000]))      # prefix of the registered SRS
        for i in range(K):
            assert np.array_equal(aff(cref, short[i]), aff(cref, cref.best_multiexp(np.ascontiguousarray(polys[i, :1000]), bases[:1000], 4)))
    finally:
        params.close()
    # unregistered bases: falls back to one general-path MSM per vector
    out = np.zeros((2, 12), dtype=np.uint64)
    two = np.ascontiguousarray(polys[:2])
    _lib.check(lib.zkhip_msm_g1_batch(two.ctypes.data, bases.ctypes.data, n, 2, out.ctypes.data))
    assert np.array_equal(aff(cref, out[1]), structured_expect(cref, polys[1], t0, d))
    a = np.ascontiguousarray(polys.copy())
    om = F.fr_encode([O.omega_for(k)])[0]
    _lib.check(lib.zkhip_ntt_fr_batch(a.ctypes.data, om.ctypes.data, k, K))
    for i in range(K):
        ref = polys[i].copy()
        cref.best_fft(ref, om, k, 4)
        assert np.array_equal(a[i], ref), i


def test_gen_walk_matches_oracle(lib, cref):
    import torch

    n = 3001
    t0, d = 0x1234567, 0xABCDEF0123456789ABCDEF
    out = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
    t0_m, d_m = F.fr_encode([t0])[0], F.fr_encode([d])[0]      # keep the arrays alive across the call
    _lib.check(lib.zkhip_g1_gen_walk_device(t0_m.ctypes.data, d_m.ctypes.data, n, out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64).reshape(n, 8)
    for i in (0, 1, 31, 32, 33, 1000, n - 1):
        exp = cref.jac_to_affine(cref.scalar_mul((t0 + i * d) % O.R_MOD, cref.generator()))
        assert np.array_equal(got[i], exp), i


def test_msm_2pow20_structured_identity_and_linearity(lib, cref):
    """BASELINE config 2 size: MSM(a, (t0 + i d) G) = [sum a_i (t0 + i d)] G, and MSM(a) + MSM(b) = MSM(a + b)."""
    n = 1 << 20
    bases, t0, d = cref.gen_bases(0x5A4B534E41500002, n)
    a = cref.gen_scalars(0x5A4B534E41500003, n, 0)
    b = cref.gen_scalars(0x5A4B534E41500004, n, 1)
    lib.zkhip_register_bases(bases.ctypes.data, n)
    try:
        ra, rb = Z.best_multiexp(a, bases), Z.best_multiexp(b, bases)
        assert np.array_equal(aff(cref, ra), structured_expect(cref, a, t0, d))
        assert np.array_equal(aff(cref, rb), structured_expect(cref, b, t0, d))
        ab = cref.field_op(1, 1, a, b)
        assert np.array_equal(aff(cref, Z.best_multiexp(ab, bases)), aff(cref, cref.jac_add(ra, rb)))
    finally:
        lib.zkhip_unregister_bases(bases.ctypes.data)


# ---------------------------------------------------------------- rows a4 / a5: best_fft and EvaluationDomain
def test_ntt_golden_vectors():
    data = json.load(open(os.path.join(GOLD, "ntt_fr.json")))
    for case in data["ntt"]:
        a = np.array([unhex(s) for s in case["input"]], dtype=np.uint64).reshape(-1, 4)
        Z.best_fft(a, np.array(unhex(case["omega"]), dtype=np.uint64), case["log_n"])
        assert a.tolist() == [unhex(s) for s in case["expected"]], case["log_n"]
    dcase = data["domain"]
    dom = Z.EvaluationDomain(dcase["j"], dcase["k"])
    H = lambda key: np.array([unhex(s) for s in dcase[key]], dtype=np.uint64).reshape(-1, 4)
    coeffs, ext = H("coeffs"), H("coeff_to_extended")
    assert np.array_equal(dom.lagrange_to_coeff(coeffs), H("lagrange_to_coeff"))
    assert np.array_equal(dom.coeff_to_extended(coeffs), ext)
    assert np.array_equal(dom.divide_by_vanishing_poly(ext), H("divide_by_vanishing_poly"))
    assert np.array_equal(dom.extended_to_coeff(ext), H("extended_to_coeff"))


@pytest.mark.parametrize("log_n", list(range(0, 21)))
def test_ntt_vs_reference_algorithm(cref, log_n):
    a = cref.gen_scalars(300 + log_n, 1 << log_n, log_n % 2)
    omega = F.fr_encode([O.omega_for(log_n)])[0]
    ref = a.copy()
    cref.best_fft(ref, omega, log_n, 8)
    Z.best_fft(a, omega, log_n)
    assert np.array_equal(a, ref)


@pytest.mark.parametrize("log_n,batch,pad", [(2, 5, 0), (6, 7, 3), (9, 4, 0), (10, 9, 5), (13, 33, 0), (15, 6, 8), (19, 3, 0)])
def test_ntt_batched_vs_reference_algorithm(lib, cref, log_n, batch, pad):
    """K polynomials in one launch set (strided layout), forward and scaled inverse, each checked against best_fft."""
    import torch

    n = 1 << log_n
    stride = n + pad
    w = O.omega_for(log_n)
    omega, omega_inv = F.fr_encode([w])[0], F.fr_encode([pow(w, -1, O.R_MOD)])[0]
    div = F.fr_encode([pow(n, -1, O.R_MOD)])[0]
    host = np.zeros((batch * stride, 4), dtype=np.uint64)
    polys = []
    for b in range(batch):
        a = cref.gen_scalars(9000 + 31 * log_n + b, n, b % 2)
        polys.append(a)
        host[b * stride:b * stride + n] = a
        host[b * stride + n:(b + 1) * stride] = 0xDEADBEEF      # padding must stay untouched
    d = torch.from_numpy(host.view(np.int64)).cuda()
    _lib.check(lib.zkhip_ntt_fr_batch_device(d.data_ptr(), omega.ctypes.data, log_n, batch, stride, None))
    torch.cuda.synchronize()
    got = d.cpu().numpy().view(np.uint64).reshape(-1, 4)
    for b in range(batch):
        ref = polys[b].copy()
        cref.best_fft(ref, omega, log_n, 4)
        assert np.array_equal(got[b * stride:b * stride + n], ref), b
        assert (got[b * stride + n:(b + 1) * stride] == 0xDEADBEEF).all()
    _lib.check(lib.zkhip_ifft_scaled_batch_device(d.data_ptr(), omega_inv.ctypes.data, log_n, div.ctypes.data, batch, stride, None))
    torch.cuda.synchronize()
    back = d.cpu().numpy().view(np.uint64).reshape(-1, 4)
    for b in range(batch):
        assert np.array_equal(back[b * stride:b * stride + n], polys[b]), b


@pytest.mark.parametrize("log_n", [22, 24])
def test_ntt_large_round_trip_and_spot_check(cref, log_n):
    """BASELINE sizes: iNTT(NTT(a)) = a limb for limb; NTT(delta_1) = powers of omega; one full compare at 2^22."""
    n = 1 << log_n
    w = O.omega_for(log_n)
    omega, omega_inv = F.fr_encode([w])[0], F.fr_encode([pow(w, -1, O.R_MOD)])[0]
    a = cref.gen_scalars(400 + log_n, n, 0)
    orig = a.copy()
    Z.best_fft(a, omega, log_n)
    if log_n == 22:
        ref = orig.copy()
        cref.best_fft(ref, omega, log_n, 16)
        assert np.array_equal(a, ref)
    dom_div = F.fr_encode([pow(n, -1, O.R_MOD)])[0]
    _lib.check(_lib.load().zkhip_ifft_scaled(a.ctypes.data, omega_inv.ctypes.data, log_n, dom_div.ctypes.data))
    assert np.array_equal(a, orig)
    delta = np.zeros((n, 4), dtype=np.uint64)
    delta[1] = F.fr_encode([1])[0]
    Z.best_fft(delta, omega, log_n)
    idx = [0, 1, 2, 12345, n // 2, n - 1]
    assert F.fr_decode(delta[idx]) == [pow(w, i, O.R_MOD) for i in idx]


@pytest.mark.parametrize("j,k", [(4, 3), (4, 7), (4, 10), (4, 13), (3, 9), (5, 8), (4, 15)])
def test_evaluation_domain_vs_reference_algorithm(cref, j, k):
    dom = Z.EvaluationDomain(j, k)
    a = cref.gen_scalars(500 + k, dom.n, 0)
    # lagrange_to_coeff = best_fft(omega_inv) then * ifft_divisor
    ref = a.copy()
    cref.best_fft(ref, dom.omega_inv, k, 4)
    cref.scale(ref, dom.ifft_divisor)
    assert np.array_equal(dom.lagrange_to_coeff(a), ref)
    # coeff_to_extended = distribute_powers_zeta(into) + zero pad + best_fft(extended_omega)
    ext = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
    ext[: dom.n] = a
    cref.distribute_powers_zeta(ext[: dom.n], dom.g_coset, dom.g_coset_inv)
    cref.best_fft(ext, dom.extended_omega, dom.extended_k, 4)
    got_ext = dom.coeff_to_extended(a)
    assert np.array_equal(got_ext, ext)
    # divide_by_vanishing_poly
    q = ext.copy()
    cref.mul_periodic(q, dom.t_evaluations)
    assert np.array_equal(dom.divide_by_vanishing_poly(got_ext), q)
    # extended_to_coeff = ifft(extended) + distribute_powers_zeta(out of) + truncate
    back = ext.copy()
    cref.best_fft(back, dom.extended_omega_inv, dom.extended_k, 4)
    cref.scale(back, dom.extended_ifft_divisor)
    cref.distribute_powers_zeta(back, dom.g_coset_inv, dom.g_coset)
    out = dom.extended_to_coeff(got_ext)
    assert out.shape[0] == dom.n * dom.quotient_poly_degree
    assert np.array_equal(out, back[: out.shape[0]])
    assert np.array_equal(out[: dom.n], a) and not out[dom.n:].any()


def test_commit_equals_evaluation_at_trapdoor(cref):
    """KZG identity on a structured SRS g[i] = s^i G with known s: commit(p) = [p(s)] G."""
    k, n = 10, 1 << 10
    g = O.SplitMix64(0x5A4B534E41500001)
    s = g.fr()
    # s^i * G = (s^i) * G: build with the C oracle's scalar multiplication
    Gp = cref.generator()
    jac = [cref.scalar_mul(pow(s, i, O.R_MOD), Gp) for i in range(n)]
    srs = np.array([cref.jac_to_affine(j) for j in jac], dtype=np.uint64)
    params = Z.ParamsKZG(k, srs)
    try:
        p = cref.gen_scalars(11, n, 0)
        coeffs = F.fr_decode(p)
        ps = 0
        for c in reversed(coeffs):
            ps = (ps * s + c) % O.R_MOD
        assert np.array_equal(aff(cref, params.commit(p)), cref.jac_to_affine(cref.scalar_mul(ps, Gp)))
    finally:
        params.close()


def test_sharded_msm_single_rank_gpu_path(cref):
    from zksnap_circuits_halo2_amd.multi_gpu import shard_range, sharded_msm

    n = 10000
    bases, t0, d = cref.gen_bases(41, n)
    sc = cref.gen_scalars(42, n, 0)
    # emulate 3 ranks sequentially on one GPU: partials + device fold
    parts = []
    for r in range(3):
        lo, hi = shard_range(n, r, 3)
        parts.append(Z.best_multiexp(np.ascontiguousarray(sc[lo:hi]), np.ascontiguousarray(bases[lo:hi])))
    out = np.zeros(12, dtype=np.uint64)
    stack = np.ascontiguousarray(np.vstack(parts))
    _lib.check(_lib.load().zkhip_g1_sum(stack.ctypes.data, 3, out.ctypes.data))
    assert np.array_equal(aff(cref, out), structured_expect(cref, sc, t0, d))
    assert np.array_equal(aff(cref, sharded_msm(sc, bases)), structured_expect(cref, sc, t0, d))


# ---------------------------------------------------------------- row a7: eval_polynomial, kate_division, batch_invert, grand product
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 1025, 32 * 32 + 1, 70000, 1 << 20])
def test_row_a7_vs_reference_algorithm(cref, n):
    from zksnap_circuits_halo2_amd import arithmetic as A

    a = cref.gen_scalars(700 + n, n, n % 2)
    x = cref.gen_scalars(800 + n, 1, 0)[0]
    assert np.array_equal(A.eval_polynomial(a, x), cref.eval_polynomial(a, x))
    assert np.array_equal(A.kate_division(a, x), cref.kate_division(a, x))
    assert np.array_equal(A.prefix_product(a), cref.prefix_product(a))
    b, ref = a.copy(), a.copy()
    A.batch_invert(b)
    cref.batch_invert(ref)
    assert np.array_equal(b, ref)
    if n >= 31:   # zeros stay zero / all zero
        assert not b[(a == 0).all(axis=1)].any()
    z = np.zeros((min(n, 100), 4), dtype=np.uint64)
    A.batch_invert(z)
    assert not z.any()
    assert np.array_equal(A.eval_polynomial(z, x), np.zeros(4, dtype=np.uint64))


@pytest.mark.parametrize("n", [4 * 65536 - 1, 4 * 65536 + 1, (1 << 19) + 3, 16 * 65536 + 7, (1 << 21) + 5])
def test_batch_invert_at_the_chunk_length_switches(cref, n):
    """`zkhip_fr_batch_invert` picks 4 ... 32 elements per thread from n (so that n / ch <= 65536) and inverts one chunk product per thread by
    division steps (csrc/fe_inverse.hpp): sizes on both sides of every switch, with zeros, +-1, 2 and small values sprinkled in, against the C
    restatement of `BatchInvert::batch_invert` (zeros stay zero) -- and x * x^-1 = 1 on a sample through Python integers."""
    from zksnap_circuits_halo2_amd import arithmetic as A

    a = cref.gen_scalars(9100 + n % 1000, n, 0)
    rng = np.random.default_rng(n)
    special = F.fr_encode([0, 1, F.R_MOD - 1, 2, F.R_MOD - 2, 3, (F.R_MOD + 1) // 2, 1 << 128, (1 << 253) + 12345])
    idx = rng.integers(0, n, size=4096)
    a[idx] = special[rng.integers(0, special.shape[0], size=idx.shape[0])]
    a[:9] = special                                      # every special value at least once, inside one chunk
    a[n - 1] = special[1]
    b, ref = a.copy(), a.copy()
    A.batch_invert(b)
    cref.batch_invert(ref)
    assert np.array_equal(b, ref)
    sample = np.concatenate([np.arange(9), idx[:40], [n - 1]])
    for x, y in zip(F.fr_decode(a[sample]), F.fr_decode(b[sample])):
        assert (x * y) % F.R_MOD == (1 if x else 0) and (x != 0 or y == 0)


def test_row_a7_golden_small():
    from zksnap_circuits_halo2_amd import arithmetic as A

    g = O.SplitMix64(2024)
    a = [g.fr() for _ in range(77)]
    x = g.fr()
    Aenc, X = F.fr_encode(a), F.fr_encode([x])[0]
    assert F.fr_decode(A.eval_polynomial(Aenc, X))[0] == O.eval_polynomial(a, x)
    assert F.fr_decode(A.kate_division(Aenc, X)) == O.kate_division(a, x)
    assert F.fr_decode(A.prefix_product(Aenc)) == O.prefix_product(a)
    B = Aenc.copy()
    A.batch_invert(B)
    assert F.fr_decode(B) == O.batch_invert(a)


@pytest.mark.parametrize("log_n,kind", [(22, 0), (22, 1), (24, 1)])
def test_msm_wrapper_sizes_structured_identity(lib, cref, log_n, kind):
    """wrapper-circuit sizes (k = 22, and k = 24 with witness-like scalars): bases generated on the device,
    MSM(a, (t0 + i d) G) = [sum a_i (t0 + i d)] G checked with the oracle's scalar arithmetic."""
    import ctypes as C

    import torch

    n = 1 << log_n
    T0, D = 0x5A4B534E41500002 + log_n, 0x9E3779B97F4A7C15F39CC0605CEDC835
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    try:
        sc = cref.gen_scalars(6000 + log_n + kind, n, kind)
        dsc = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
        exp = cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, T0, D), cref.generator()))
        assert np.array_equal(aff(cref, out.cpu().numpy().view(np.uint64)), exp)
    finally:
        _lib.check(lib.zkhip_release_bases(h))


def test_concurrent_host_calls_are_serialised_correctly(cref):
    """best_multiexp / best_fft called from several host threads at once (rayon-style callers): the library serialises
    them internally; every result must still be right."""
    import threading

    n = 3000
    bases, t0, d = cref.gen_bases(71, n)
    results, errors = {}, []

    def work(tid):
        try:
            for rep in range(4):
                sc = cref.gen_scalars(7100 + 10 * tid + rep, n, rep % 2)
                got = aff(cref, Z.best_multiexp(sc, bases))
                ok = np.array_equal(got, structured_expect(cref, sc, t0, d))
                a = cref.gen_scalars(7200 + 10 * tid + rep, 1 << 10, 0)
                ref = a.copy()
                om = F.fr_encode([O.omega_for(10)])[0]
                cref.best_fft(ref, om, 10, 1)
                Z.best_fft(a, om, 10)
                results[(tid, rep)] = ok and np.array_equal(a, ref)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(results) == 16 and all(results.values())


# ---------------------------------------------------------------- C++ host mirror (include/zkhip.hpp)
def test_cpp_host_mirror(cref, tmp_path):
    """The compiled-host mirror of the reference interface (EvaluationDomain, ParamsKZG, best_fft, ...) end to end."""
    import struct
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "tests", "cpp", "host_mirror_driver")
    assert os.path.exists(drv), "build it with __graft_entry__.build()"
    j, k = 4, 11
    n = 1 << k
    poly = cref.gen_scalars(4242, n, 0)
    g, t0, d = cref.gen_bases(4243, n)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<II", j, k))
        f.write(poly.tobytes())
        f.write(g.tobytes())
    subprocess.check_call([drv, str(fin), str(fout)], timeout=300)
    raw = np.fromfile(fout, dtype=np.uint64)
    pos = 0

    def take(count, width):
        nonlocal pos
        out = raw[pos:pos + count * width].reshape(count, width)
        pos += count * width
        return out

    dom = O.EvaluationDomain(j, k)
    assert F.fr_decode(take(1, 4))[0] == dom.omega and F.fr_decode(take(1, 4))[0] == dom.extended_omega
    assert np.array_equal(aff(cref, take(1, 12)[0]), structured_expect(cref, poly, t0, d))
    enc1 = lambda v: F.fr_encode([v])[0]
    ref = poly.copy(); cref.best_fft(ref, enc1(dom.omega), k, 4)
    assert np.array_equal(take(n, 4), ref)
    ref = poly.copy(); cref.best_fft(ref, enc1(dom.omega_inv), k, 4); cref.scale(ref, enc1(dom.ifft_divisor))
    assert np.array_equal(take(n, 4), ref)
    ext = np.zeros((dom.extended_len(), 4), dtype=np.uint64); ext[:n] = poly
    cref.distribute_powers_zeta(ext[:n], enc1(dom.g_coset), enc1(dom.g_coset_inv)); cref.best_fft(ext, enc1(dom.extended_omega), dom.extended_k, 4)
    assert np.array_equal(take(dom.extended_len(), 4), ext)
    q = ext.copy(); cref.mul_periodic(q, F.fr_encode(dom.t_evaluations))
    assert np.array_equal(take(dom.extended_len(), 4), q)
    back = take(3 * n, 4)
    assert np.array_equal(back[:n], poly) and not back[n:].any()
    assert np.array_equal(take(1, 4)[0], cref.eval_polynomial(poly, enc1(dom.omega)))
    assert np.array_equal(take(n - 1, 4), cref.kate_division(poly, enc1(dom.omega)))
    # section 8(f) pieces of the mirror
    pv = F.fr_decode(poly)
    assert F.fr_decode(take(n, 4)) == O.grand_product(pv, pv[1:] + pv[:1])
    lin = [(int(poly[i, 0]) >> 7) % 37 for i in range(n)]
    exp_in, exp_tab = O.permute_expression_pair(lin, [i % 37 for i in range(n)], n - 6)
    assert F.fr_decode(take(n - 6, 4)) == exp_in
    assert F.fr_decode(take(n - 6, 4)) == exp_tab
    assert F.fr_decode(take(n, 4)) == [(pv[i] * pv[(i + 1) % n] + 7 * pv[i]) % O.R_MOD for i in range(n)]
    # ParamsKZG::setup(k, s): g[i] = [s^i] G; commit(p) = [p(s)] G; commit_lagrange(e) = [interpolant of e at s] G
    s_trap = 0x1234567
    g4 = take(4, 8)
    assert [O.affine_from_limbs([int(x) for x in row]) for row in g4] == O.structured_srs(s_trap, 4)
    c1, c2 = take(1, 12)[0], take(1, 12)[0]
    assert F.g1_decode_jacobian(c1) == O.scalar_mul(O.eval_polynomial(pv, s_trap), O.G1_GEN)
    assert F.g1_decode_jacobian(c2) == O.scalar_mul(O.eval_polynomial(dom.lagrange_to_coeff(pv), s_trap), O.G1_GEN)
    # ParamsKZG::write -> read -> write gives identical bytes of the RawBytes size; the re-read SRS commits alike; truncation throws
    assert int(take(1, 1)[0, 0]) == 1
    assert F.g1_decode_jacobian(take(1, 12)[0]) == F.g1_decode_jacobian(c2)
    assert int(take(1, 1)[0, 0]) == 1
    assert int(take(1, 1)[0, 0]) == 1                        # g_to_lagrange(setup.g) == setup.g_lagrange
    # permutation_products: 3 columns (rotations of the polynomial's values) in sets of 2, chained through z[n - 6]
    vals = [pv[c + 1:] + pv[:c + 1] for c in range(3)]
    sig = [pv[5 * (c + 1):] + pv[:5 * (c + 1)] for c in range(3)]
    beta, gamma, usable, last = dom.omega, 12345, n - 6, 1
    for lo in (0, 2):
        z = [last]
        for i in range(n - 1):
            if i < usable:
                a = b = 1
                for c in range(lo, min(lo + 2, 3)):
                    a = a * (vals[c][i] + pow(O.FR_DELTA, c, O.R_MOD) * beta % O.R_MOD * pow(dom.omega, i, O.R_MOD) + gamma) % O.R_MOD
                    b = b * (vals[c][i] + beta * sig[c][i] + gamma) % O.R_MOD
                z.append(z[-1] * a % O.R_MOD * pow(b, -1, O.R_MOD) % O.R_MOD)
            else:
                z.append(z[-1])
        last = z[usable]
        assert F.fr_decode(take(n, 4)) == z
    assert pos == raw.size


def _gather_fold_worker(rank, world, port, n, q, rank_width=12):
    """one of `world` processes sharing the single GPU of the test box; gloo stands in for RCCL"""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from oracle import cpu_ref as Cr
    from zksnap_circuits_halo2_amd import _lib as L
    from zksnap_circuits_halo2_amd.multi_gpu import gather_fold_device, shard_range

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = L.load()
    bases, t0, d = Cr.gen_bases(4321, n)
    sc = Cr.gen_scalars(8765, n, 0)
    lo, hi = shard_range(n, rank, world)
    d_b = torch.from_numpy(np.ascontiguousarray(bases[lo:hi]).view(np.int64)).cuda()
    d_s = torch.from_numpy(np.ascontiguousarray(sc[lo:hi]).view(np.int64)).cuda()
    width = 12 if rank_width == 12 else 16          # bench.py exchanges the bare 96-byte points; 128-byte slots are accepted too
    d_out = torch.zeros(width, dtype=torch.int64, device="cuda")
    d_gather = torch.zeros(width * world, dtype=torch.int64, device="cuda")
    d_final = torch.zeros(width, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    h = C.c_uint64(0)
    L.check(lib.zkhip_prepare_bases_device(d_b.data_ptr(), hi - lo, C.byref(h)))
    for _ in range(2):      # twice: the buffers are reused step after step in bench.py
        L.check(lib.zkhip_msm_g1_prepared_device(h, 0, d_s.data_ptr(), hi - lo, d_out.data_ptr(), stream))
        gather_fold_device(d_out, d_gather, d_final, stream)
    torch.cuda.synchronize()
    exp = Cr.jac_to_affine(Cr.scalar_mul(Cr.expected_scalar(sc, t0, d), Cr.generator()))
    got = Cr.jac_to_affine(np.ascontiguousarray(d_final.cpu().numpy().view(np.uint64)[:12]))
    q.put((rank, bool(np.array_equal(got, exp))))
    lib.zkhip_release_bases(h)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width", [12, 16])
def test_bench_exchange_path_two_ranks_on_one_gpu(width):
    """the N > 1 step of bench.py (prepared MSM on the rank's shard -> gather the 96-byte partials -> fold) with two processes on this
    box's single GPU and gloo in place of RCCL"""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 5001
    procs = [ctx.Process(target=_gather_fold_worker, args=(r, world, port, n, q, width)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_device_buffer_pipeline_without_torch(lib, cref):
    """a host that does not link HIP: zkhip_alloc / upload / download around the `_device` entry points (stream = NULL), chained
    iNTT -> commit -> extended coset NTT -> back, compared with the host-buffer entry points"""
    import ctypes as C

    k, ek = 10, 12
    n = 1 << k
    a = cref.gen_scalars(4242, n, 0)
    bases, t0, d = cref.gen_bases(4243, n)
    d_a, d_b, d_out = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.zkhip_alloc(n * 32, C.byref(d_a)))
    _lib.check(lib.zkhip_alloc(n * 64, C.byref(d_b)))
    _lib.check(lib.zkhip_alloc(96, C.byref(d_out)))
    try:
        _lib.check(lib.zkhip_upload(d_a, a.ctypes.data, n * 32))
        _lib.check(lib.zkhip_upload(d_b, bases.ctypes.data, n * 64))
        om_i = F.fr_encode([pow(F.omega_for(k), -1, F.R_MOD)])[0]
        div = F.fr_encode([pow(n, -1, F.R_MOD)])[0]
        _lib.check(lib.zkhip_ifft_scaled_device(d_a, om_i.ctypes.data, k, div.ctypes.data, None))        # lagrange_to_coeff
        _lib.check(lib.zkhip_msm_g1_device(d_a, d_b, n, d_out, None))                                     # commit
        coeffs = np.zeros((n, 4), dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        _lib.check(lib.zkhip_download(coeffs.ctypes.data, d_a, n * 32))
        _lib.check(lib.zkhip_download(out.ctypes.data, d_out, 96))
        _lib.check(lib.zkhip_sync())
        ref = a.copy()
        _lib.check(lib.zkhip_ifft_scaled(ref.ctypes.data, om_i.ctypes.data, k, div.ctypes.data))
        assert np.array_equal(coeffs, ref)
        assert np.array_equal(aff(cref, out), aff(cref, Z.best_multiexp(ref, bases)))
    finally:
        for p in (d_a, d_b, d_out):
            _lib.check(lib.zkhip_free(p))
    assert lib.zkhip_free(None) == 0
    z = C.c_void_p(1)
    assert lib.zkhip_alloc(0, C.byref(z)) == 0 and not z.value
    assert lib.zkhip_alloc(1 << 50, C.byref(z)) == -4 and not z.value            # ZKHIP_ENOMEM, no abort


@pytest.mark.parametrize("k,batch", [(3, 1), (10, 3), (13, 2)])
def test_coset_transforms_device_batch_vs_host_entry_points(lib, cref, k, batch):
    """zkhip_coeff_to_extended_device / zkhip_extended_to_coeff_device (batched, strided, device-resident) against the host-buffer
    entry points and the reference algorithm"""
    import torch

    dom = Z.EvaluationDomain(4, k)
    n, en, ek = dom.n, dom.extended_len(), dom.extended_k
    a_stride, e_stride, o_stride = n + 8, en + 16, 3 * n + 4
    polys = [cref.gen_scalars(9100 + k + b, n, b % 2) for b in range(batch)]
    d_a = torch.zeros(batch * a_stride * 4, dtype=torch.int64, device="cuda")
    for b, p in enumerate(polys):
        d_a[b * a_stride * 4:(b * a_stride + n) * 4] = torch.from_numpy(p.view(np.int64).reshape(-1)).cuda()
    d_e = torch.zeros(batch * e_stride * 4, dtype=torch.int64, device="cuda")
    d_o = torch.zeros(batch * o_stride * 4, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.zkhip_coeff_to_extended_device(d_a.data_ptr(), a_stride, k, d_e.data_ptr(), e_stride, ek, batch,
                                                  dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data, s))
    _lib.check(lib.zkhip_extended_to_coeff_device(d_e.data_ptr(), e_stride, ek, dom.extended_omega_inv.ctypes.data, dom.extended_ifft_divisor.ctypes.data,
                                                  dom.g_coset.ctypes.data, d_o.data_ptr(), o_stride, 3 * n, batch, s))
    torch.cuda.synchronize()
    ext = d_e.cpu().numpy().view(np.uint64).reshape(batch, e_stride, 4)
    back = d_o.cpu().numpy().view(np.uint64).reshape(batch, o_stride, 4)
    enc1 = lambda v: F.fr_encode([v])[0]
    od = O.EvaluationDomain(4, k)
    for b, p in enumerate(polys):
        assert np.array_equal(ext[b, :en], dom.coeff_to_extended(p)), b
        ref = np.zeros((en, 4), dtype=np.uint64); ref[:n] = p
        cref.distribute_powers_zeta(ref[:n], enc1(od.g_coset), enc1(od.g_coset_inv)); cref.best_fft(ref, enc1(od.extended_omega), ek, 4)
        assert np.array_equal(ext[b, :en], ref), b
        assert not ext[b, en:].any() and not back[b, 3 * n:].any()            # the padding between polynomials is untouched
        assert np.array_equal(back[b, :n], p) and not back[b, n:3 * n].any(), b


@pytest.mark.parametrize("pattern", ["all_equal", "two_values", "ragged_zero_tail"])
def test_msm_wide_window_path_under_skew(lib, cref, pattern):
    """the wide-window configuration (prepared bases, c = 20, two-level sort: the library's choice from n = 2^20) with scalar vectors
    that put every entry of a window into one or two buckets, and with a ragged length whose tail is all zero"""
    import ctypes as C

    import torch

    n = (1 << 20) + (37 if pattern == "ragged_zero_tail" else 0)
    T0, D = 0x5A4B534E41500777, 0x9E3779B97F4A7C15F39CC0605CEDC835
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    assert lib.zkhip_prepared_window_bits(h) == 20                 # the library's own choice from 2^20
    try:
        if pattern == "all_equal":
            sc = np.tile(cref.gen_scalars(1, 1, 0), (n, 1))
        elif pattern == "two_values":
            two = cref.gen_scalars(2, 2, 0)
            sc = two[np.arange(n) % 2]
        else:
            sc = cref.gen_scalars(3, n, 1)
            sc[n // 2:] = 0
        sc = np.ascontiguousarray(sc)
        dsc = torch.from_numpy(sc.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
        exp = cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, T0, D), cref.generator()))
        assert np.array_equal(aff(cref, out.cpu().numpy().view(np.uint64)), exp)
    finally:
        _lib.check(lib.zkhip_release_bases(h))


def test_params_kzg_setup_small_vs_oracle(lib):
    """ParamsKZG::setup with a known trapdoor: g[i] = [s^i] G and g_lagrange[i] = [L_i(s)] G against the big-int oracle"""
    k, s = 4, 0x1234567890ABCDEF1234567890ABCDEF
    n = 1 << k
    params = Z.ParamsKZG.setup(k, s)
    try:
        srs = O.structured_srs(s, n)
        assert [O.affine_from_limbs([int(x) for x in row]) for row in params.g] == srs
        w = O.omega_for(k)
        mult = (pow(s, n, O.R_MOD) - 1) * pow(n, -1, O.R_MOD) % O.R_MOD
        for i in range(n):
            li = mult * pow(w, i, O.R_MOD) % O.R_MOD * pow((s - pow(w, i, O.R_MOD)) % O.R_MOD, -1, O.R_MOD) % O.R_MOD
            assert O.affine_from_limbs([int(x) for x in params.g_lagrange[i]]) == O.scalar_mul(li, O.G1_GEN), i
    finally:
        params.close()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 7])
def test_g1_fft_matches_the_scalar_transform(lib, log_n):
    """best_fft::<G1>: for points a_j G the transform is [NTT(a)_i] G -- the scalar transform comes from the big-int oracle, the
    points from its scalar multiplication; input includes the identity and a repeated point"""
    import ctypes as C

    n = 1 << log_n
    rng = random.Random(90 + log_n)
    a = [rng.randrange(O.R_MOD) for _ in range(n)]
    if n >= 4:
        a[1] = 0                                              # the identity among the inputs
        a[3] = a[2]                                           # equal points meet in a butterfly (doubling inside the addition)
    omega = O.omega_for(log_n)
    pts = [O.to_jac(O.scalar_mul(v, O.G1_GEN)) if v else O.JAC_ID for v in a]
    buf = np.array([[lim for c in P for lim in O.limbs4(O.to_mont(c, O.Q_MOD))] for P in pts], dtype=np.uint64).reshape(n, 12)
    d = C.c_void_p()
    _lib.check(lib.zkhip_alloc(n * 96, C.byref(d)))
    try:
        _lib.check(lib.zkhip_upload(d, buf.ctypes.data, n * 96))
        om = F.fr_encode([omega])[0]
        _lib.check(lib.zkhip_g1_fft_device(d, om.ctypes.data, log_n, None))
        out = np.zeros((n, 12), dtype=np.uint64)
        _lib.check(lib.zkhip_download(out.ctypes.data, d, n * 96))
    finally:
        lib.zkhip_free(d)
    expect = O.best_fft(a, omega, log_n)
    for i in range(n):
        assert F.g1_decode_jacobian(out[i]) == (O.scalar_mul(expect[i], O.G1_GEN) if expect[i] else None), i


@pytest.mark.parametrize("k", [3, 9])
def test_g_to_lagrange_agrees_with_the_trapdoor_formula(lib, cref, k):
    """two routes to the Lagrange-basis SRS: setup's closed form [L_i(s)] G (fixed-base multiplications) and g_to_lagrange's inverse
    FFT over the points of g -- different kernels, same points; then from_parts commits like setup"""
    from zksnap_circuits_halo2_amd import kzg

    s = 0x6C61_6772_616E_6765_0000_0000_0000_0001_0203
    params = Z.ParamsKZG.setup(k, s)
    try:
        assert np.array_equal(kzg.g_to_lagrange(params.g, k), params.g_lagrange)
        # the host-buffer entry point with the reference function's memory (`Vec<G1>` in, `Vec<G1Affine>` out: rust-shim/commitment_patch.rs item 5):
        # Jacobian input with z = 1 and, for every third point, another representative of the same point (x z^2, y z^3, z)
        n = 1 << k
        one_q = np.array(O.limbs4(O.to_mont(1, O.Q_MOD)), dtype=np.uint64)
        jac = np.concatenate([params.g, np.broadcast_to(one_q, (n, 4))], axis=1)
        for i in range(0, n, 3):
            pt = O.affine_from_limbs([int(v) for v in params.g[i]])
            z = 0x1234567 + i
            jac[i] = np.array(O.limbs4(O.to_mont(pt[0] * z * z % O.Q_MOD, O.Q_MOD)) + O.limbs4(O.to_mont(pt[1] * z * z * z % O.Q_MOD, O.Q_MOD))
                              + O.limbs4(O.to_mont(z, O.Q_MOD)), dtype=np.uint64)
        jac = np.ascontiguousarray(jac)
        out = np.zeros((n, 8), dtype=np.uint64)
        _lib.check(lib.zkhip_g_to_lagrange(jac.ctypes.data, k, out.ctypes.data))
        assert np.array_equal(out, params.g_lagrange)
        evals = cref.gen_scalars(4242 + k, 1 << k, 0)
        expected = aff(cref, params.commit_lagrange(evals))
    finally:
        params.close()
    derived = Z.ParamsKZG.from_parts(k, params.g, None, params.g2, params.s_g2)
    try:
        assert np.array_equal(aff(cref, derived.commit_lagrange(evals)), expected)
    finally:
        derived.close()


def test_g1_fft_bad_arguments(lib):
    import ctypes as C

    om = F.fr_encode([1])[0]
    d = C.c_void_p()
    _lib.check(lib.zkhip_alloc(96, C.byref(d)))
    try:
        assert lib.zkhip_g1_fft_device(None, om.ctypes.data, 0, None) == -1
        assert lib.zkhip_g1_fft_device(d, om.ctypes.data, 27, None) == -1
        assert lib.zkhip_g_to_lagrange_device(d, 0, d, None) == -1          # aliasing
        assert lib.zkhip_g_to_lagrange_device(None, 0, d, None) == -1
    finally:
        lib.zkhip_free(d)


def test_params_kzg_write_read_round_trip(lib, cref, tmp_path):
    """ParamsKZG::write / read (RawBytes layout, srs.py): a file written from a set-up SRS reads back to the same tables, G2 points
    included, and the re-read parameters commit to the same point (their tables are registered afresh)"""
    from zksnap_circuits_halo2_amd import srs as S

    k, s = 9, 0x0BADC0DE_5EED_0001_0002_0003
    params = Z.ParamsKZG.setup(k, s)
    path = tmp_path / f"kzg_bn254_{k}.srs"
    try:
        assert S.g2_decode(params.get_g2()) == S.G2_GENERATOR and S.g2_decode(params.get_s_g2()) == S.g2_mul(s)
        with open(path, "wb") as f:
            params.write(f)
        assert path.stat().st_size == 4 + 2 * (1 << k) * 64 + 256
        poly = cref.gen_scalars(77, 1 << k, 0)
        c_before = aff(cref, params.commit(poly)), aff(cref, params.commit_lagrange(poly))
    finally:
        params.close()
    with open(path, "rb") as f:
        again = Z.ParamsKZG.read(f)
    try:
        assert again.k == k and np.array_equal(again.g, params.g) and np.array_equal(again.g_lagrange, params.g_lagrange)
        assert np.array_equal(again.g2, params.g2) and np.array_equal(again.s_g2, params.s_g2)
        c_after = aff(cref, again.commit(poly)), aff(cref, again.commit_lagrange(poly))
        assert all(np.array_equal(a, b) for a, b in zip(c_before, c_after))
    finally:
        again.close()


@pytest.mark.parametrize("k", [10, 15])
def test_kzg_commitments_agree_across_bases(lib, cref, k):
    """a structured SRS ties the MSM, the NTT and the setup together: for evaluations e of a polynomial p over the domain,
    commit_lagrange(e) = commit(lagrange_to_coeff(e)) = [p(s)] G"""
    from zksnap_circuits_halo2_amd import arithmetic as A

    s = 0x0123456789ABCDEF_FEDCBA9876543210_0F1E2D3C4B5A6978 % O.R_MOD
    n = 1 << k
    params = Z.ParamsKZG.setup(k, s)
    try:
        dom = Z.EvaluationDomain(4, k)
        evals = cref.gen_scalars(31337 + k, n, 0)
        coeffs = dom.lagrange_to_coeff(evals)
        c1 = aff(cref, params.commit_lagrange(evals))
        c2 = aff(cref, params.commit(coeffs))
        p_at_s = F.fr_decode(A.eval_polynomial(coeffs, F.fr_encode([s])[0]))[0]
        exp = cref.jac_to_affine(cref.scalar_mul(p_at_s, cref.generator()))
        assert np.array_equal(c1, c2) and np.array_equal(c1, exp)
    finally:
        params.close()


def test_c_abi_rejects_bad_arguments_without_aborting(lib):
    """every entry point returns ZKHIP_EINVAL (-1) with a message on null / inconsistent arguments (the reference's functions are
    infallible, the C ABI never aborts: SURVEY.md section 8(b))"""
    import ctypes as C

    buf = np.zeros((64, 4), dtype=np.uint64)
    pts = np.zeros((64, 8), dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    om = F.fr_encode([F.omega_for(4)])[0]
    p, q, o, w = buf.ctypes.data, pts.ctypes.data, out.ctypes.data, om.ctypes.data
    bad = [
        lib.zkhip_msm_g1(None, q, 4, o), lib.zkhip_msm_g1(p, None, 4, o), lib.zkhip_msm_g1(p, q, 4, None),
        lib.zkhip_ntt_fr(None, w, 4), lib.zkhip_ntt_fr(p, None, 4), lib.zkhip_ntt_fr(p, w, 29),
        lib.zkhip_ifft_scaled(p, w, 4, None), lib.zkhip_mul_periodic(p, 16, None, 4), lib.zkhip_mul_periodic(p, 16, p, 0),
        lib.zkhip_coeff_to_extended(p, 5, p, 4, w, w), lib.zkhip_extended_to_coeff(p, 29, w, w, w, p, 4),
        lib.zkhip_coeff_to_extended_device(p, 8, 4, p, 64, 6, 1, w, w, None),          # stride shorter than the polynomial
        lib.zkhip_extended_to_coeff_device(p, 64, 6, w, w, w, p, 8, 16, 1, None),      # out_stride < out_len
        lib.zkhip_fr_eval_polynomial(None, 4, w, o), lib.zkhip_fr_kate_division(p, 4, None, p), lib.zkhip_fr_batch_invert(None, 4),
        lib.zkhip_fr_prefix_product(p, 4, None), lib.zkhip_fr_grand_product(p, None, 4, p), lib.zkhip_lookup_permute(p, p, 4, None, p),
        lib.zkhip_fr_eval_polynomial_batch_device(None, 2, 4, w, p, None), lib.zkhip_fr_eval_rows(None, None, 0, 2, 0, p),
        lib.zkhip_register_bases(None, 4), lib.zkhip_unregister_bases(q), lib.zkhip_release_bases(123456),
        lib.zkhip_msm_g1_prepared_device(123456, 0, p, 4, o, None), lib.zkhip_prepare_bases_device(None, 4, C.byref(C.c_uint64())),
        lib.zkhip_g1_fixed_base_mul_device(None, 4, q, None), lib.zkhip_g1_sum(None, 2, o), lib.zkhip_upload(None, p, 8), lib.zkhip_download(p, None, 8),
        lib.zkhip_alloc(8, None), lib.zkhip_test_field_op(2, 0, p, p, p, 4), lib.zkhip_test_g1_op(9, q, q, o, 1),
    ]
    assert bad == [-1] * len(bad), bad
    assert lib.zkhip_last_error()                                          # a message is available
    assert lib.zkhip_prepared_window_bits(123456) == -1
    # and the library is still usable afterwards
    a = F.fr_encode(list(range(16)))
    _lib.check(lib.zkhip_ntt_fr(a.ctypes.data, w, 4))
    assert F.fr_decode(a) == O.best_fft(list(range(16)), F.omega_for(4), 4)


def test_ntt_batch_larger_than_the_scratch_cap_runs_in_sub_batches(lib):
    """5 polynomials of 2^24 (2.7 GB) exceed the 2 GiB scratch cap: the batch is transformed in sub-batches of 4 + 1; each polynomial must
    equal its single-call transform, and the inverse batch must restore the input"""
    import torch

    L, B = 24, 5
    n = 1 << L
    x = torch.randint(0, 1 << 62, (B, n, 4), dtype=torch.int64, device="cuda")
    x[:, :, 3] &= (1 << 61) - 1
    orig = x.clone()
    om = F.fr_encode([F.omega_for(L)])[0]
    omi = F.fr_encode([pow(F.omega_for(L), -1, F.R_MOD)])[0]
    div = F.fr_encode([pow(n, -1, F.R_MOD)])[0]
    single = orig[4].clone()
    _lib.check(lib.zkhip_ntt_fr_device(single.data_ptr(), om.ctypes.data, L, None))
    _lib.check(lib.zkhip_ntt_fr_batch_device(x.data_ptr(), om.ctypes.data, L, B, n, None))
    torch.cuda.synchronize()
    assert torch.equal(x[4], single)
    _lib.check(lib.zkhip_ifft_scaled_batch_device(x.data_ptr(), omi.ctypes.data, L, div.ctypes.data, B, n, None))
    torch.cuda.synchronize()
    assert torch.equal(x, orig)


@pytest.mark.parametrize("log_n", [25, 26, 28])
def test_ntt_above_the_direct_twiddle_table_limit(lib, log_n):
    """2^25 .. 2^28 (the ABI's maximum): pass twiddles come from two sqrt-size tables instead of one direct table.  Device-resident:
    NTT(delta_j)[i] = omega^(i j) at sampled i, and iNTT(NTT(a)) = a on random data."""
    import torch

    n = 1 << log_n
    w = O.omega_for(log_n)
    om, omi = F.fr_encode([w])[0], F.fr_encode([pow(w, -1, O.R_MOD)])[0]
    div = F.fr_encode([pow(n, -1, O.R_MOD)])[0]
    j = 0x2F3A5 % n
    d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    d[j] = torch.from_numpy(F.fr_encode([1]).view(np.int64))[0].cuda()
    _lib.check(lib.zkhip_ntt_fr_device(d.data_ptr(), om.ctypes.data, log_n, None))
    torch.cuda.synchronize()
    idx = [0, 1, 2, 4097, 65537, n // 2 + 3, n - 1]
    got = F.fr_decode(d[idx].cpu().numpy().view(np.uint64))
    assert got == [pow(w, i * j, O.R_MOD) for i in idx]
    del d
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda")
    a[:, 3] = torch.randint(0, 1 << 61, (n,), dtype=torch.int64, device="cuda")
    orig = a.clone()
    _lib.check(lib.zkhip_ntt_fr_device(a.data_ptr(), om.ctypes.data, log_n, None))
    _lib.check(lib.zkhip_ifft_scaled_device(a.data_ptr(), omi.ctypes.data, log_n, div.ctypes.data, None))
    torch.cuda.synchronize()
    assert torch.equal(a, orig)


def test_shutdown_and_lazy_reinit(lib, cref):
    """zkhip_shutdown releases every device resource (tables, scratch, cached twiddles, the generator table); the next call
    initialises lazily again and gives the same results"""
    n = 3000
    bases, t0, d = cref.gen_bases(99, n)
    sc = cref.gen_scalars(98, n, 0)
    before = aff(cref, Z.best_multiexp(sc, bases))
    a = cref.gen_scalars(97, 1 << 10, 0)
    f1 = a.copy(); Z.best_fft(f1, F.fr_encode([F.omega_for(10)])[0], 10)
    params = Z.ParamsKZG.setup(6, 12345)
    g_before = params.g.copy()
    params.close()
    lib.zkhip_shutdown()
    lib.zkhip_shutdown()                                   # idempotent
    assert np.array_equal(aff(cref, Z.best_multiexp(sc, bases)), before)
    f2 = a.copy(); Z.best_fft(f2, F.fr_encode([F.omega_for(10)])[0], 10)
    assert np.array_equal(f1, f2)
    params = Z.ParamsKZG.setup(6, 12345)
    assert np.array_equal(params.g, g_before)
    params.close()
    assert np.array_equal(before, structured_expect(cref, sc, t0, d))


def test_ntt_eight_elements_per_thread_variant(cref, tmp_path):
    """the NTT kernel has two instances (4 elements per thread: default; 8: ZKHIP_NTT_ELEMS=8, read once per process): run the second one
    in a child process over one-, two- and three-pass sizes with input / output scales and compare with the oracle"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ntt8.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import zksnap_circuits_halo2_amd as Z\n"
        "from oracle import cpu_ref as Cr\n"
        "from zksnap_circuits_halo2_amd import fields as F\n"
        "for L in (3, 6, 9, 10, 13, 17, 19):\n"
        "    a = Cr.gen_scalars(900 + L, 1 << L, L % 2); om = F.fr_encode([F.omega_for(L)])[0]\n"
        "    ref = a.copy(); Cr.best_fft(ref, om, L, 4); Z.best_fft(a, om, L)\n"
        "    assert np.array_equal(a, ref), L\n"
        "dom = Z.EvaluationDomain(4, 11)\n"
        "p = Cr.gen_scalars(77, dom.n, 0)\n"
        "back = dom.extended_to_coeff(dom.coeff_to_extended(p))\n"
        "assert np.array_equal(back[:dom.n], p) and not back[dom.n:].any()\n"
        "print('ok')\n")
    env = dict(os.environ, ZKHIP_NTT_ELEMS="8")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("prepared", [False, True])
def test_msm_opposite_points_cancel_inside_buckets(lib, cref, prepared):
    """P and -P with the same scalar meet in the same bucket of every window: the accumulator passes through the identity in the middle of a
    task (XYZZ: ZZ = 0) and must pick up the next point as a fresh start; the MSM of the cancelling pairs is the identity, and with one
    extra (scalar, point) it is that single product"""
    n = 4096
    bases, t0, d = cref.gen_bases(51, n // 2)
    neg = bases.copy()
    neg[:, 4:8] = cref.field_op(0, 2, np.zeros((n // 2, 4), dtype=np.uint64), np.ascontiguousarray(bases[:, 4:8]))      # y -> -y
    pts = np.ascontiguousarray(np.stack([bases, neg], axis=1).reshape(n, 8))                                                # P0, -P0, P1, -P1, ...
    sc_half = cref.gen_scalars(52, n // 2, 0)
    sc_half[: n // 4] = sc_half[0]                                                                                          # a quarter of the pairs share one scalar: long tasks
    sc = np.ascontiguousarray(np.repeat(sc_half, 2, axis=0))
    if prepared:
        _lib.check(lib.zkhip_register_bases(pts.ctypes.data, n))
    try:
        assert F.g1_decode_jacobian(Z.best_multiexp(sc, pts)) is None
        sc2 = sc.copy()
        sc2[-1] = 0                                       # the last -P drops out: what is left is s * P_last
        got = aff(cref, Z.best_multiexp(sc2, pts))
        k = F.fr_decode(sc[-2:-1])[0]
        assert np.array_equal(got, cref.jac_to_affine(cref.scalar_mul(k, pts[-2])))
    finally:
        if prepared:
            _lib.check(lib.zkhip_unregister_bases(pts.ctypes.data))


def test_params_downsize_equals_smaller_setup(lib):
    """`ParamsKZG::downsize(k)`: the first 2^k points of g and the Lagrange basis re-derived from them (g_to_lagrange on the GPU) are
    exactly what setup produces for the smaller k with the same trapdoor"""
    with Z.ParamsKZG.setup(8, 4242) as big, Z.ParamsKZG.setup(6, 4242) as small:
        with big.downsize(6) as down:
            assert down.k == 6 and np.array_equal(down.g, small.g) and np.array_equal(down.g_lagrange, small.g_lagrange)
            assert np.array_equal(down.s_g2, small.s_g2)
        with pytest.raises(ValueError):
            big.downsize(9)
