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


def test_g2_compression_round_trip_and_flags():
    """`G2Affine::{to_bytes, from_bytes}` on the host (the two G2 points of a SerdeFormat::Processed parameter file): both flag layouts,
    identity, the sign bit selects between y and -y, non-canonical / off-curve / mis-flagged encodings are refused"""
    rng = random.Random(9)
    pts = [srs.G2_GENERATOR, srs.g2_mul(0xABCDEF), srs.g2_mul(rng.randrange(O.R_MOD)), srs.g2_mul(O.R_MOD - 1), None]
    for lay in (0, 1):
        for P in pts:
            b = srs.g2_compress(P, lay)
            assert len(b) == 64 and srs.g2_decompress(b, lay) == P
        P = pts[1]
        b = bytearray(srs.g2_compress(P, lay))
        b[63] ^= 0x40 if lay == 0 else 0x80                                         # flip the sign flag: the other root
        Q = srs.g2_decompress(bytes(b), lay)
        assert Q == (P[0], ((-P[1][0]) % F.Q_MOD, (-P[1][1]) % F.Q_MOD))
    # x and the sign as the encoding defines them: little-endian canonical c0 || c1, sign = lsb of y.c0
    P = pts[2]
    b = srs.g2_compress(P, 0)
    assert int.from_bytes(b[:32], "little") == P[0][0] and int.from_bytes(bytes(b[32:63]) + bytes([b[63] & 0x3F]), "little") == P[0][1]
    assert (b[63] >> 6) & 1 == P[1][0] & 1 and b[63] >> 7 == 0
    assert srs.g2_compress(None, 0)[63] == 0x80 and srs.g2_compress(None, 1) == bytes(64)
    with pytest.raises(ValueError):                                                  # x.c0 = q: not canonical
        srs.g2_decompress(F.Q_MOD.to_bytes(32, "little") + bytes(32), 0)
    with pytest.raises(ValueError):                                                  # identity flag on a non-zero x
        srs.g2_decompress(bytes([1]) + bytes(62) + bytes([0x80]), 0)
    bad = next(x for x in range(1, 50) if srs._f2sqrt(tuple((c + d) % F.Q_MOD for c, d in zip(srs._f2mul(srs._f2mul((x, 0), (x, 0)), (x, 0)), srs.G2_B))) is None)
    with pytest.raises(ValueError):                                                  # x^3 + b' is not a square in Fq2
        srs.g2_decompress(bad.to_bytes(32, "little") + bytes(32), 0)


def test_fq2_square_roots():
    rng = random.Random(10)
    squares = 0
    for _ in range(60):
        a = (rng.randrange(F.Q_MOD), rng.randrange(F.Q_MOD))
        s2 = srs._f2mul(a, a)
        r = srs._f2sqrt(s2)
        assert r is not None and srs._f2mul(r, r) == s2
        r = srs._f2sqrt(a)
        if r is not None:
            squares += 1
            assert srs._f2mul(r, r) == a
    assert 10 < squares < 50                                                        # about half of Fq2 are squares
    for a0 in (4, F.Q_MOD - 4, 0, 3, F.Q_MOD - 3):                                  # c1 = 0: the root is real or purely imaginary
        r = srs._f2sqrt((a0, 0))
        assert r is not None and srs._f2mul(r, r) == (a0 % F.Q_MOD, 0)
