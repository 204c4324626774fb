"""GPU (-m gpu): `SerdeFormat::Processed` -- compressed G1 points (`G1Affine::{to_bytes, from_bytes}` of halo2curves [DEP], one call per
point in `ParamsKZG::{write_custom, read_custom}` and the key readers / writers) batched on the device (csrc/serde.hip), against the
encoding computed with big integers: x canonical little-endian, sign = lsb of canonical y, identity flag; both flag layouts; every way an
encoding can be invalid; and the parameter / key files written and read back in that format."""
import ctypes as C
import io
import random

import numpy as np
import pytest

import zksnap_circuits_halo2_amd as Z
from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F, keygen as KG, srs as S

pytestmark = pytest.mark.gpu
Q = O.Q_MOD


def expected_bytes(P, layout):
    if P is None:
        b = bytearray(32)
        if layout == 0:
            b[31] = 0x80
        return bytes(b)
    b = bytearray(P[0].to_bytes(32, "little"))
    b[31] |= (P[1] & 1) << (6 if layout == 0 else 7)
    return bytes(b)


def some_points(n, seed):
    rng = random.Random(seed)
    pts = [O.G1_GEN, O.scalar_mul(2, O.G1_GEN), O.scalar_mul(O.R_MOD - 1, O.G1_GEN), None]
    acc = O.scalar_mul(rng.randrange(O.R_MOD), O.G1_GEN)
    step = O.scalar_mul(rng.randrange(O.R_MOD), O.G1_GEN)
    while len(pts) < n:
        pts.append(acc)
        acc = O.add(acc, step)
    return pts[:n]


@pytest.mark.parametrize("layout", [0, 1])
def test_g1_compress_and_decompress_vs_big_integers(lib, layout):
    pts = some_points(300, 11 + layout)
    enc = F.g1_encode(pts)
    data = S.g1_compress(enc, layout)
    assert data == b"".join(expected_bytes(P, layout) for P in pts)
    back = S.g1_decompress(data, len(pts), layout)
    assert np.array_equal(back, enc)                                               # canonical Montgomery limbs, (0, 0) for the identity
    # the sign flag selects the root: flipping it gives -P
    flipped = bytearray(data)
    for i in range(len(pts)):
        if pts[i] is not None:
            flipped[32 * i + 31] ^= 0x40 if layout == 0 else 0x80
    neg = S.g1_decompress(bytes(flipped), len(pts), layout)
    assert np.array_equal(neg, F.g1_encode([None if P is None else (P[0], Q - P[1]) for P in pts]))


def test_g1_decompress_refuses_invalid_encodings(lib):
    pts = some_points(64, 5)
    good = bytearray(S.g1_compress(F.g1_encode(pts), 0))
    n = len(pts)

    def first_bad(buf, layout=0):
        src = np.frombuffer(bytes(buf), dtype=np.uint8)
        out = np.zeros((n, 8), dtype=np.uint64)
        bad = C.c_uint64(0)
        _lib.check(lib.zkhip_g1_decompress(src.ctypes.data, n, out.ctypes.data, layout, C.byref(bad)))
        return bad.value, out

    assert first_bad(good)[0] == n
    # x not the abscissa of a curve point (x^3 + 3 a non-residue)
    x_off = next(x for x in range(2, 100) if pow((x ** 3 + 3) % Q, (Q - 1) // 2, Q) != 1)
    b = bytearray(good); b[32 * 9:32 * 10] = x_off.to_bytes(32, "little")
    idx, out = first_bad(b)
    assert idx == 9 and not out[9].any()                                           # the refused slot is left as (0, 0)
    # x = q: not canonical (its low 254 bits would be a valid residue otherwise)
    b = bytearray(good); b[32 * 20:32 * 21] = Q.to_bytes(32, "little")
    assert first_bad(b)[0] == 20
    # identity flag on a non-zero x; identity flag with the sign bit
    b = bytearray(good); b[32 * 30 + 31] |= 0x80
    assert first_bad(b)[0] == 30
    b = bytearray(good); b[32 * 3:32 * 4] = bytes(31) + bytes([0xC0])              # slot 3 is the identity in `some_points`
    assert first_bad(b)[0] == 3
    # two bad slots: the smaller index is reported
    b = bytearray(good); b[32 * 50:32 * 51] = Q.to_bytes(32, "little"); b[32 * 12:32 * 13] = x_off.to_bytes(32, "little")
    assert first_bad(b)[0] == 12
    with pytest.raises(ValueError):
        S.g1_decompress(bytes(b), n, 0)
    # x = 0 without the identity flag: 3 is not a square mod q
    b = bytearray(good); b[32 * 5:32 * 6] = bytes(32)
    assert first_bad(b)[0] == 5
    # argument checks
    bad = C.c_uint64(0)
    assert lib.zkhip_g1_decompress(None, 4, None, 0, C.byref(bad)) == -1            # ZKHIP_EINVAL
    assert lib.zkhip_g1_compress(None, 0, None, 0) == 0
    one = np.zeros(8, dtype=np.uint64)
    assert lib.zkhip_g1_compress(one.ctypes.data, 1, one.ctypes.data, 7) == -1


def test_g1_compress_device_entry_points_and_a_large_table(lib, cref):
    """2^18 points of a generated walk through the `_device` entry points: compress -> decompress is the identity, spot checks against
    big integers"""
    n = 1 << 18
    t0, d = F.fr_encode([1234577])[0], F.fr_encode([991])[0]
    ptrs = [C.c_void_p() for _ in range(3)]
    for p, size in zip(ptrs, (n * 64, n * 32, n * 64)):
        _lib.check(lib.zkhip_alloc(size, C.byref(p)))
    try:
        _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, d.ctypes.data, n, ptrs[0], None))
        _lib.check(lib.zkhip_g1_compress_device(ptrs[0], n, ptrs[1], 0, None))
        bad = C.c_uint64(0)
        _lib.check(lib.zkhip_g1_decompress_device(ptrs[1], n, ptrs[2], 0, C.byref(bad), None))
        assert bad.value == n
        a, b = np.zeros((n, 8), dtype=np.uint64), np.zeros((n, 8), dtype=np.uint64)
        comp = np.zeros(n * 32, dtype=np.uint8)
        _lib.check(lib.zkhip_download(a.ctypes.data, ptrs[0], n * 64))
        _lib.check(lib.zkhip_download(b.ctypes.data, ptrs[2], n * 64))
        _lib.check(lib.zkhip_download(comp.ctypes.data, ptrs[1], n * 32))
        assert np.array_equal(a, b)
        inv = pow(F.MONT, -1, Q)
        for i in (0, 1, 77777, n - 1):
            x = sum(int(a[i, j]) << (64 * j) for j in range(4)) * inv % Q
            y = sum(int(a[i, 4 + j]) << (64 * j) for j in range(4)) * inv % Q
            assert comp[32 * i:32 * i + 32].tobytes() == expected_bytes((x, y), 0)
    finally:
        for p in ptrs:
            lib.zkhip_free(p)


def test_params_kzg_processed_round_trip(lib, cref, tmp_path):
    k, s = 8, 0x5EED_0042
    with Z.ParamsKZG.setup(k, s) as params:
        buf = io.BytesIO()
        params.write_custom(buf, S.PROCESSED)
        g, gl, g2, s_g2 = params.g.copy(), params.g_lagrange.copy(), params.g2.copy(), params.s_g2.copy()
    data = buf.getvalue()
    assert len(data) == 4 + 2 * (1 << k) * 32 + 128
    assert data[4:36] == expected_bytes(O.G1_GEN, 0)                               # g[0] = G
    assert data[-128:-64] == S.g2_compress(S.G2_GENERATOR) and data[-64:] == S.g2_compress(S.g2_mul(s))
    with Z.ParamsKZG.read_custom(io.BytesIO(data), S.PROCESSED) as again:
        assert again.k == k and np.array_equal(again.g, g) and np.array_equal(again.g_lagrange, gl)
        assert np.array_equal(again.g2, g2) and np.array_equal(again.s_g2, s_g2)
        poly = cref.gen_scalars(78, 1 << k, 0)
        assert np.array_equal(cref.jac_to_affine(again.commit(poly)), cref.jac_to_affine(cref.best_multiexp(poly, g, 2)))
    with pytest.raises(ValueError):
        Z.ParamsKZG.read_custom(io.BytesIO(data[:-5]), S.PROCESSED)
    corrupt = bytearray(data); corrupt[4 + 32 * 17:4 + 32 * 18] = Q.to_bytes(32, "little")
    with pytest.raises(ValueError):
        Z.ParamsKZG.read_custom(io.BytesIO(bytes(corrupt)), S.PROCESSED)
    # the legacy flag layout is a different file that reads back to the same parameters
    with Z.ParamsKZG.read_custom(io.BytesIO(data), S.PROCESSED) as p0:
        legacy = io.BytesIO()
        p0.write_custom(legacy, S.PROCESSED, flag_layout=1)
    assert legacy.getvalue() != data
    with Z.ParamsKZG.read_custom(io.BytesIO(legacy.getvalue()), S.PROCESSED, flag_layout=1) as p1:
        assert np.array_equal(p1.g, g) and np.array_equal(p1.s_g2, s_g2)


def test_proving_key_processed_round_trip(lib):
    from tests.test_gpu_prover_flow import BLIND, K, N, PERM_COLUMNS, enc, toy_circuit
    from zksnap_circuits_halo2_amd import evaluation as E

    rng = random.Random(21)
    circ = toy_circuit(rng)
    fixed = [enc(c) for c in circ["fixed"]]
    cs = E.ConstraintSystem(num_fixed=3, num_advice=2, permutation_columns=PERM_COLUMNS, blinding_factors=BLIND, degree=4)
    asm = KG.Assembly(N, 3)
    for c in [(0, 1, 2, 2), (1, 10, 1, 20), (0, 13, 1, 30)]:
        asm.copy(*c)
    with Z.ParamsKZG.setup(K, 0xFACE) as params:
        vk = KG.keygen_vk(params, cs, fixed, asm)
        pk = KG.keygen_pk(params, vk, cs, fixed, asm)
    raw, proc = io.BytesIO(), io.BytesIO()
    pk.write(raw, KG.RAW_BYTES_UNCHECKED)
    pk.write(proc, KG.PROCESSED)
    assert len(proc.getvalue()) == len(raw.getvalue()) - 32 * (3 + 3)              # six commitments, 32 bytes shorter each
    # scalars are canonical integers in the Processed file: the first fixed value
    off = 8 + 6 * 32 + 3 * (4 + 4 * N * 32) + 4 + 4                                # vk | l0, l_last, l_active_row | count | length
    assert int.from_bytes(proc.getvalue()[off:off + 32], "little") == circ["fixed"][0][0] % O.R_MOD
    back = KG.ProvingKey.read(io.BytesIO(proc.getvalue()), KG.PROCESSED, cs)
    again = io.BytesIO()
    back.write(again, KG.RAW_BYTES_UNCHECKED)
    assert again.getvalue() == raw.getvalue()
    bad = bytearray(proc.getvalue()); bad[off:off + 32] = O.R_MOD.to_bytes(32, "little")     # a scalar that is not canonical
    with pytest.raises(ValueError):
        KG.ProvingKey.read(io.BytesIO(bytes(bad)), KG.PROCESSED, cs)
