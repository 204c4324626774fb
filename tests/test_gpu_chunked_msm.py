"""GPU: the chunked host-buffer MSM (round 4).  `zkhip_msm_g1` on a registered SRS -- the call `ParamsKZG::commit` makes under an
unmodified `create_proof` (/root/reference/aggregator/src/wrapper.rs:129) -- cuts scalar vectors of >= 2^21 elements into pieces of ~2^20
that cross PCIe on a copy stream while the pieces before them are sorted and accumulated into ONE shared bucket set (csrc/msm.hip
msm_chunk_add / msm_chunk_finish, csrc/capi.hip msm_shard_enqueue_chunked).  A sum over points is a sum over pieces of sums, so the
result must be the same group element as the structured-SRS identity for every size, offset, piece count and scalar distribution."""
import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F

pytestmark = pytest.mark.gpu
D = 0x9E3779B97F4A7C15F39CC0605CEDC835
T0 = 0x5A4B534E41500002 + 4004


def walk_host(lib, n):
    import torch

    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    torch.cuda.synchronize()
    return np.ascontiguousarray(bases.cpu().numpy().view(np.uint64).reshape(n, 8))


def expect(cref, sc, t0):
    return cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, t0, D), cref.generator()))


def msm(lib, sc, bases):
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.zkhip_msm_g1(sc.ctypes.data, bases.ctypes.data, sc.shape[0], out.ctypes.data))
    return out


@pytest.fixture(scope="module")
def srs(lib):
    n = (1 << 22) + 77
    bases = walk_host(lib, n)
    _lib.check(lib.zkhip_register_bases(bases.ctypes.data, n))
    yield bases
    _lib.check(lib.zkhip_unregister_bases(bases.ctypes.data))


@pytest.mark.parametrize("lo,n", [(0, 1 << 21), (0, (1 << 21) + 5), (3, (1 << 21) - 1 + 4), (1000, 3 * (1 << 20) + 7), (0, (1 << 22) + 77), (70, 1 << 22)])
def test_chunked_commit_equals_structured_identity(lib, cref, srs, lo, n):
    """2, 3 and 4 pieces (ragged: the pieces differ by one scalar), at offsets into the registered array"""
    sc = cref.gen_scalars(7000 + n % 97 + lo, n, 0)
    got = cref.jac_to_affine(msm(lib, sc, srs[lo:lo + n]))
    assert np.array_equal(got, expect(cref, sc, (T0 + lo * D) % O.R_MOD))


def test_chunked_commit_with_skewed_columns(lib, cref, srs):
    """witness-like columns: a selector (half ones, half zeros -- a heavy bucket in every piece), a column of equal values, a column whose second
    piece is all zero (an empty piece must leave the bucket set untouched), and -1 everywhere (the negated table points)"""
    n = 3 * (1 << 20) + 7
    one = F.fr_encode([1])[0]
    minus_one = F.fr_encode([O.R_MOD - 1])[0]
    cases = {}
    sel = np.zeros((n, 4), dtype=np.uint64); sel[::2] = one
    cases["selector"] = sel
    cases["equal"] = np.ascontiguousarray(np.broadcast_to(F.fr_encode([0x1234567890ABCDEF1122334455667788])[0], (n, 4)))
    holes = cref.gen_scalars(7100, n, 0); holes[n // 3 + 1: 2 * (n // 3) + 1] = 0
    cases["empty_piece"] = holes
    cases["minus_one"] = np.ascontiguousarray(np.broadcast_to(minus_one, (n, 4)))
    for name, sc in cases.items():
        sc = np.ascontiguousarray(sc)
        got = cref.jac_to_affine(msm(lib, sc, srs[:n]))
        assert np.array_equal(got, expect(cref, sc, T0)), name


def test_chunked_pieces_on_virtual_shards(lib, cref):
    """two shards of 2^21 + 1 points on the one card: each shard's MSM is chunked, and the second shard's pieces cross PCIe (into their own region of
    the lane's scalar buffer) while the first shard is still being accumulated"""
    n = (1 << 22) + 2
    bases = walk_host(lib, n)
    sc = cref.gen_scalars(7200, n, 0)
    _lib.check(lib.zkhip_set_msm_shards(2))
    try:
        _lib.check(lib.zkhip_register_bases(bases.ctypes.data, n))
        try:
            for _ in range(2):
                assert np.array_equal(cref.jac_to_affine(msm(lib, sc, bases)), expect(cref, sc, T0))
        finally:
            _lib.check(lib.zkhip_unregister_bases(bases.ctypes.data))
    finally:
        lib.zkhip_set_msm_shards(0)
