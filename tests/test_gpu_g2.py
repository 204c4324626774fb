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
    a, b = enc(Pa), enc(Pb)
    dbl = lambda P: O.g2_add(P, P)
    for op, f in ((3, lambda P, Q: O.g2_add(dbl(P), Q)), (4, lambda P, Q: dbl(dbl(P)))):
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
