"""GPU (-m gpu): `best_multiexp::<G2Affine>` -- Fq2 / twist-curve arithmetic (csrc/ec2.hpp) and the G2 MSM (csrc/msm_g2.hip) against
the oracle's big-integer G2 (oracle/bn254.py, pinned to the EIP-197 generator, curve membership and the group order).  The north star
names MSM on G1/G2; the reference prover has no G2 MSM call site (/root/reference/aggregator/src/wrapper.rs:1142-1144 only reads g2)."""
import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, arithmetic as A, fields as F

pytestmark = pytest.mark.gpu


def enc(points):
    return np.array([O.g2_affine_to_limbs(P) for P in points], dtype=np.uint64).reshape(len(points), 16)


def dec(row):
    return O.g2_jac_from_limbs([int(x) for x in row])


def walk(t0, d, n):
    """[(t0 + i d) G2 for i < n]: one scalar multiplication, then a walk of Jacobian additions, normalised at the end"""
    P, D = O.g2_to_jac(O.g2_scalar_mul(t0, O.G2_GEN)), O.g2_to_jac(O.g2_scalar_mul(d, O.G2_GEN))
    out = []
    for _ in range(n):
        out.append(P)
        P = O.g2_jac_add(P, D)
    return [O.g2_to_affine(J) for J in out]


def test_g2_point_ops_vs_oracle(lib):
    g = O.SplitMix64(77)
    n = 48
    Pa = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in range(n)]
    Pb = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in range(n)]
    Pa[3] = None; Pb[4] = None; Pa[5] = None; Pb[5] = None                     # identities on either / both sides
    Pb[6] = Pa[6]; Pb[7] = Pa[7]                                             # doubling through the general (even row) and the mixed (odd row) addition
    Pb[8] = O.g2_neg(Pa[8]); Pb[9] = O.g2_neg(Pa[9])                         # opposite points, both formulas
    a, b = enc(Pa), enc(Pb)
    for op, f in ((0, lambda P, Q: O.g2_add(P, Q)), (1, lambda P, Q: O.g2_add(P, P)), (2, lambda P, Q: O.g2_add(P, O.g2_neg(Q)))):
        out = np.zeros((n, 24), dtype=np.uint64)
        _lib.check(lib.zkhip_test_g2_op(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n))
        assert [dec(r) for r in out] == [f(P, Q) for P, Q in zip(Pa, Pb)], op


def test_g2_quad_cooperative_point_ops_vs_oracle(lib):
    """the quad formulas (csrc/ec2_quad.hpp: four lanes share every Fq2 product) that the G2 MSM's combine / pyramid / window-sum / fold
    kernels use: 2 a + b and 4 a against the oracle, incl. identities, 2 a = b (the doubling case inside the addition) and 2 a = -b"""
    g = O.SplitMix64(78)
    n = 40
    Pa = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in range(n)]
    Pb = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in range(n)]
    Pa[3] = None; Pb[4] = None; Pa[5] = None; Pb[5] = None
    Pb[6] = O.g2_add(Pa[6], Pa[6])                                            # 2 a + b with b = 2 a: equal points
    Pb[7] = O.g2_neg(O.g2_add(Pa[7], Pa[7]))                                  # b = -2 a: the sum is the identity
    dbl = lambda P: O.g2_add(P, P)
    Pb[8] = dbl(dbl(Pa[8]))                                                   # op 5: 4 a + b with b = 4 a (equal points inside the lazy addition)
    Pb[9] = O.g2_neg(dbl(dbl(Pa[9])))                                         # op 5: b = -4 a
    a, b = enc(Pa), enc(Pb)
    for op, f in ((3, lambda P, Q: O.g2_add(dbl(P), Q)), (4, lambda P, Q: dbl(dbl(P))),
                  (5, lambda P, Q: O.g2_add(dbl(dbl(P)), Q)), (6, lambda P, Q: dbl(dbl(dbl(dbl(P)))))):     # 5 / 6: the lazy quad chains of the fold
        out = np.zeros((n, 24), dtype=np.uint64)
        _lib.check(lib.zkhip_test_g2_op(op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n))
        assert [dec(r) for r in out] == [f(P, Q) for P, Q in zip(Pa, Pb)], op


@pytest.mark.parametrize("n", [0, 1, 2, 3, 33, 257])
def test_g2_msm_vs_naive_oracle(lib, n):
    g = O.SplitMix64(500 + n)
    pts = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in range(n)]
    sc = [g.fr() for _ in range(n)]
    if n >= 33:
        pts[5] = None                                   # identity base with a non-zero scalar
        sc[6] = 0; sc[7] = 1; sc[8] = O.R_MOD - 1; sc[9] = 1 << 253
        pts[11] = pts[10]; sc[11] = sc[10]              # repeated (base, scalar): a doubling inside a bucket
        pts[13] = O.g2_neg(pts[12]); sc[13] = sc[12]    # opposite points with equal scalars cancel inside a bucket
    got = dec(A.best_multiexp_g2(F.fr_encode(sc), enc(pts)))
    assert got == O.g2_msm_naive(sc, pts)


@pytest.mark.parametrize("n,kind", [(5000, 0), (20000, 1)])
def test_g2_msm_structured_identity(lib, cref, n, kind):
    """MSM(a, (t0 + i d) G2) = [sum a_i (t0 + i d)] G2 on a walk of n points: windows of 10 - 12 bits, thousands of buckets, tasks"""
    t0, d = 0x5A4B534E41500002 + n, 0x9E3779B97F4A7C15F39CC0605CEDC835
    pts = walk(t0, d, n)
    sc = cref.gen_scalars(8800 + n, n, kind)
    got = dec(A.best_multiexp_g2(sc, enc(pts)))
    assert got == O.g2_scalar_mul(cref.expected_scalar(sc, t0, d), O.G2_GEN)


def test_g2_msm_device_resident_matches_host_api(lib, cref):
    import torch

    n = 3000
    pts = enc(walk(12345, 678, n))
    sc = cref.gen_scalars(8900, n, 0)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    d_bs = torch.from_numpy(pts.view(np.int64)).cuda()
    out = torch.zeros(24, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_msm_g2_device(d_sc.data_ptr(), d_bs.data_ptr(), n, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert dec(out.cpu().numpy().view(np.uint64)) == dec(A.best_multiexp_g2(sc, pts))


# ---------------------------------------------------------------------------------------------------------------------------------
# the sizes the bench times (2^16: windows of 16 bits, 2^15 buckets per window, tasks of 64) and one ragged size above, on a walk that the
# DEVICE extends: 256 host-made points + a per-block shift, added elementwise by the G2 addition kernel, normalised on the host with one
# shared inversion (the big-integer oracle would need minutes to walk 2^17 points and invert every z)
def _f2_batch_inv(zs):
    pref, run = [], (1, 0)
    for z in zs:
        pref.append(run)
        run = O.f2_mul(run, z)
    inv = O.f2_inv(run)
    out = [None] * len(zs)
    for i in range(len(zs) - 1, -1, -1):
        out[i] = O.f2_mul(inv, pref[i])
        inv = O.f2_mul(inv, zs[i])
    return out


def device_walk(lib, t0, d, n, block=256):
    """limbs ((n, 16) uint64) of [(t0 + i d) G2 for i < n]"""
    assert n % block == 0
    base = walk(t0, d, block)
    step = O.g2_scalar_mul(block * d % O.R_MOD, O.G2_GEN)
    shifts, S = [], None
    for _ in range(n // block):
        shifts.append(S)
        S = O.g2_add(S, step)
    a = np.tile(enc(base), (n // block, 1))
    b = np.repeat(enc(shifts), block, axis=0)
    out = np.zeros((n, 24), dtype=np.uint64)
    _lib.check(lib.zkhip_test_g2_op(0, a.ctypes.data, b.ctypes.data, out.ctypes.data, n))
    def jac_of(row):                                                # (X, Y, Z) as Fq2 triples, not normalised (no point of a walk is the identity)
        v = [O.from_mont(O.from_limbs([int(x) for x in row[4 * c:4 * c + 4]]), O.Q_MOD) for c in range(6)]
        return (v[0], v[1]), (v[2], v[3]), (v[4], v[5])

    jac = [jac_of(r) for r in out]
    assert all(J[2] != (0, 0) for J in jac)
    zi = _f2_batch_inv([J[2] for J in jac])                          # one inversion for all the z
    pts = []
    for (X, Y, Z), iz in zip(jac, zi):
        iz2 = O.f2_mul(iz, iz)
        pts.append((O.f2_mul(X, iz2), O.f2_mul(Y, O.f2_mul(iz2, iz))))
    assert O.g2_on_curve(pts[0]) and O.g2_on_curve(pts[-1]) and pts[block + 1] == O.g2_scalar_mul((t0 + (block + 1) * d) % O.R_MOD, O.G2_GEN)
    return enc(pts)


@pytest.fixture(scope="module")
def g2_walk_2p17(lib):
    t0, d = 0x5A4B534E41500002 + 171717, 0x9E3779B97F4A7C15F39CC0605CEDC835
    n = (1 << 17) + 256
    return t0, d, device_walk(lib, t0, d, n)


@pytest.mark.parametrize("n,kind", [(1 << 16, 0), ((1 << 17) + 3, 1)])
def test_g2_msm_at_the_bench_shape(lib, cref, g2_walk_2p17, n, kind):
    """2^16 uniform scalars (what bench.py times as msm_g2_2^16) and 2^17 + 3 witness-like scalars: MSM(a, (t0 + i d) G2) = [sum a_i (t0 + i d)] G2"""
    t0, d, pts = g2_walk_2p17
    sc = cref.gen_scalars(8850 + kind, n, kind)
    got = dec(A.best_multiexp_g2(sc, np.ascontiguousarray(pts[:n])))
    assert got == O.g2_scalar_mul(cref.expected_scalar(sc, t0, d), O.G2_GEN)


@pytest.mark.parametrize("case", ["all_equal", "selector", "ones_and_minus_ones", "one_bucket_per_window"])
def test_g2_msm_skewed_scalars(lib, cref, g2_walk_2p17, case):
    """the skew cases of the G1 suite: a column of equal values / of selector bits puts a large share of all entries into one bucket of every
    window (heavy buckets, cooperative task records); 1 and -1 cancel pairwise inside buckets"""
    t0, d, pts = g2_walk_2p17
    n = 1 << 14
    one, minus_one = F.fr_encode([1])[0], F.fr_encode([O.R_MOD - 1])[0]
    sc = np.zeros((n, 4), dtype=np.uint64)
    if case == "all_equal":
        sc[:] = F.fr_encode([0x1234567890ABCDEF1234567890ABCDEF])[0]
    elif case == "selector":
        sc[::2] = one
    elif case == "ones_and_minus_ones":
        sc[::2] = one; sc[1::2] = minus_one
    else:
        sc[:] = F.fr_encode([sum(5 << (16 * w) for w in range(16))])[0]          # digit 5 in every 16-bit window
    got = dec(A.best_multiexp_g2(sc, np.ascontiguousarray(pts[:n])))
    assert got == O.g2_scalar_mul(cref.expected_scalar(sc, t0, d), O.G2_GEN)


def test_g2_msm_glv_edge_scalars_vs_naive_oracle(lib):
    """the GLV split of the G2 path (round 5) at the scalars where a decomposition goes wrong first: 0, 1, r - 1, lambda and its neighbours,
    lambda^2, the lattice vectors' entries, powers of two around 2^127 / 2^128, the largest 254-bit value -- against the oracle's naive sum"""
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    a1, nb1, a2 = 0x89d3256894d213e3, 0x6f4d8248eeb859fc8211bbeb7d4f1128, 0x6f4d8248eeb859fd0be4e1541221250b
    r = O.R_MOD
    vals = [0, 1, 2, r - 1, r - 2, lam, lam + 1, lam - 1, r - lam, lam * lam % r, (lam * lam + 1) % r, a1, nb1, a2, r - a1, r - a2,
            1 << 127, (1 << 127) - 1, (1 << 128) - 1, 1 << 128, (1 << 253) + 12345, r >> 1, (r >> 1) + 1, 0xFFFF, 0x10000, 0x8000, 0x7FFF]
    vals = [v % r for v in vals]
    g = O.SplitMix64(79)
    pts = [O.g2_scalar_mul(g.fr(), O.G2_GEN) for _ in vals]
    got = dec(A.best_multiexp_g2(F.fr_encode(vals), enc(pts)))
    assert got == O.g2_msm_naive(vals, pts)


def test_g2_msm_two_level_sort_size(lib, cref, g2_walk_2p17):
    """2^18 + 5 points: from 2^18 the GLV digits take the two-level LDS sort (msm.hip msm_two_level) with 288-byte work points.  Bases = the 2^17
    walk repeated, so the expected scalar is the sum over the two halves"""
    t0, d, pts = g2_walk_2p17
    half = 1 << 17
    n = 2 * half + 5
    bases = np.ascontiguousarray(np.concatenate([pts[:half], pts[:half], pts[:5]]))
    sc = cref.gen_scalars(8870, n, 0)
    want = (cref.expected_scalar(np.ascontiguousarray(sc[:half]), t0, d) + cref.expected_scalar(np.ascontiguousarray(sc[half:2 * half]), t0, d)
            + cref.expected_scalar(np.ascontiguousarray(sc[2 * half:]), t0, d)) % O.R_MOD
    assert dec(A.best_multiexp_g2(sc, bases)) == O.g2_scalar_mul(want, O.G2_GEN)
