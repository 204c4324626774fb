"""CPU: the SRS file layer (zksnap_circuits_halo2_amd/srs.py) -- RawBytes layout, truncation / corruption errors, and the host G2
arithmetic behind `ParamsKZG::setup`'s s_g2, checked against the oracle's independent G1 code where a shared structure exists
(group laws) and against the curve equation / subgroup order otherwise."""
import io  # the default reader checks every point on the GPU; these CPU tests use the host-side sampling mode (check_points=64)
import random
import struct

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import fields as F, srs


def small_params(k=3, s=0x1234567):
    n = 1 << k
    g = F.g1_encode(O.structured_srs(s, n))
    gl = F.g1_encode([O.scalar_mul(7 + i, O.G1_GEN) for i in range(n)])
    return k, g, gl, srs.g2_encode(srs.G2_GENERATOR), srs.g2_encode(srs.g2_mul(s))


def test_g2_generator_and_group_law():
    G = srs.G2_GENERATOR
    assert srs.g2_is_on_curve(G)
    assert srs.g2_mul(O.R_MOD - 1) == (G[0], ((-G[1][0]) % F.Q_MOD, (-G[1][1]) % F.Q_MOD))      # (r - 1) G = -G: the order is r
    assert srs.g2_add(srs.g2_mul(O.R_MOD - 1), G) is None
    rng = random.Random(5)
    for _ in range(4):
        a, b = rng.randrange(O.R_MOD), rng.randrange(O.R_MOD)
        P = srs.g2_add(srs.g2_mul(a), srs.g2_mul(b))
        assert P == srs.g2_mul(a + b) and srs.g2_is_on_curve(P)
    assert srs.g2_mul(0) is None and srs.g2_mul(O.R_MOD) is None
    assert srs.g2_mul(2) == srs.g2_add(G, G)


def test_g2_memory_round_trip():
    P = srs.g2_mul(0xDEADBEEFCAFE)
    enc = srs.g2_encode(P)
    assert enc.shape == (16,) and srs.g2_decode(enc) == P
    assert srs.g2_decode(srs.g2_encode(None)) is None
    # Montgomery form: the first coordinate limb group is x.c0 * 2^256 mod q
    m = sum(int(enc[j]) << (64 * j) for j in range(4))
    assert m == P[0][0] * F.MONT % F.Q_MOD
    bad = enc.copy(); bad[3] = 0xFFFFFFFFFFFFFFFF
    with pytest.raises(ValueError):
        srs.g2_decode(bad)


def test_raw_bytes_layout_and_round_trip():
    k, g, gl, g2, s_g2 = small_params()
    blob = srs.params_to_bytes(k, g, gl, g2, s_g2)
    n = 1 << k
    assert len(blob) == 4 + 2 * n * 64 + 256
    assert struct.unpack("<I", blob[:4])[0] == k
    assert blob[4:4 + 64] == g[0].astype("<u8").tobytes()                    # the memory of G1Affine, untouched
    assert blob[4 + n * 64:4 + n * 64 + 64] == gl[0].astype("<u8").tobytes()
    assert blob[-256:-128] == g2.astype("<u8").tobytes() and blob[-128:] == s_g2.astype("<u8").tobytes()
    k2, ga, gla, g2a, s_g2a = srs.read_params(io.BytesIO(blob), check_points=64)
    assert k2 == k and np.array_equal(ga, g) and np.array_equal(gla, gl) and np.array_equal(g2a, g2) and np.array_equal(s_g2a, s_g2)
    assert ga.dtype == np.uint64 and ga.flags["C_CONTIGUOUS"]


@pytest.mark.parametrize("cut", [0, 3, 4 + 64, 4 + 8 * 64 + 10, 4 + 16 * 64, 4 + 16 * 64 + 255])
def test_truncated_file_is_reported(cut):
    blob = srs.params_to_bytes(*small_params())
    with pytest.raises(ValueError, match="truncated"):
        srs.read_params(io.BytesIO(blob[:cut]), check_points=64)


def test_corruption_is_reported():
    k, g, gl, g2, s_g2 = small_params()
    blob = bytearray(srs.params_to_bytes(k, g, gl, g2, s_g2))
    bad = bytearray(blob); bad[0:4] = struct.pack("<I", 77)
    with pytest.raises(ValueError, match="k = 77"):
        srs.read_params(io.BytesIO(bytes(bad)), check_points=64)
    bad = bytearray(blob); bad[4 + 64 * 2 + 5] ^= 1                          # one bit of g[2].x
    with pytest.raises(ValueError, match="point 2"):
        srs.read_params(io.BytesIO(bytes(bad)), check_points=64)
    srs.read_params(io.BytesIO(bytes(bad)), check_points=0)                  # RawBytesUnchecked: accepted as is
    bad = bytearray(blob); bad[-100] ^= 1
    with pytest.raises(ValueError, match="s_g2"):
        srs.read_params(io.BytesIO(bytes(bad)), check_points=64)
    with pytest.raises(ValueError):
        srs.write_params(io.BytesIO(), k + 1, g, gl, g2, s_g2)


def test_identity_points_are_accepted():
    k, g, gl, g2, s_g2 = small_params()
    g[1] = 0                                                                 # (0, 0) = the identity in G1Affine memory
    out = srs.read_params(io.BytesIO(srs.params_to_bytes(k, g, gl, g2, s_g2)), check_points=64)
    assert not out[1][1].any()
