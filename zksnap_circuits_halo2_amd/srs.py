"""`ParamsKZG::{write, read}` file format and the G2 half of `ParamsKZG::setup`.

[DEP] halo2-axiom poly/kzg/commitment.rs `write_custom` / `read_custom` with `SerdeFormat::RawBytes` (what `write` / `read` use), reached
from `gen_srs` (/root/reference/aggregator/benches/wrapper_circuit.rs:35,49,69, /root/reference/aggregator/src/wrapper.rs:961) and from the
browser worker that fetches `kzg_bn254_{k}.srs` (/root/reference/voter/frontend/app/worker.js:218-225).  The dependency is un-vendored, so
the layout below is the published one restated, not checked against a reference file (parity unpinned):

    k            u32 little-endian
    g            2^k   x 64 B   G1Affine, x || y, each 4 x u64 LE limbs in Montgomery form -- the memory of `&[G1Affine]`
    g_lagrange   2^k   x 64 B   the same
    g2           128 B          G2Affine, x.c0 || x.c1 || y.c0 || y.c1, Montgomery limbs
    s_g2         128 B          the same

Raw bytes are the in-memory representation, which is also what the C ABI takes: a file maps straight onto `zkhip_register_bases`.
`SerdeFormat::Processed` (`write_custom` / `read_custom`) stores the same sequence with every point compressed (G1: 32 B, G2: 64 B; x
canonical little-endian, flags in the top bits of the last byte): the G1 tables are compressed / decompressed on the GPU
(`zkhip_g1_compress` / `zkhip_g1_decompress`: one square root per point), the two G2 points on the host.  The flag layout differs between
halo2curves releases and no reference file pins it: `flag_layout` 0 = sign in bit 6, identity in bit 7 (>= 0.3.2), 1 = sign in bit 7.
G2 never reaches the GPU (the prover only carries g2 / s_g2 into the verifying key: /root/reference/aggregator/src/wrapper.rs:1143-1144);
its arithmetic here is host big-int code for the single multiplication [s]G2 that `setup` needs."""
from __future__ import annotations

import io
import struct
from typing import BinaryIO, Optional, Tuple

import numpy as np

from .fields import MONT, Q_MOD, R_MOD

Fq2 = Tuple[int, int]
# generator of the order-r subgroup of the sextic twist y^2 = x^3 + 3 / (9 + i)  (the alt_bn128 G2 generator, as in halo2curves bn256::G2)
G2_GENERATOR = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def _f2mul(a: Fq2, b: Fq2) -> Fq2:
    return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def _f2sub(a: Fq2, b: Fq2) -> Fq2:
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def _f2inv(a: Fq2) -> Fq2:
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q_MOD)
    return (a[0] * d % Q_MOD, -a[1] * d % Q_MOD)


G2_B = _f2mul((3, 0), _f2inv((9, 1)))


def g2_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    x3 = _f2mul(_f2mul(x, x), x)
    return _f2mul(y, y) == ((x3[0] + G2_B[0]) % Q_MOD, (x3[1] + G2_B[1]) % Q_MOD)


def g2_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if ((P[1][0] + Q[1][0]) % Q_MOD, (P[1][1] + Q[1][1]) % Q_MOD) == (0, 0):
            return None
        lam = _f2mul(_f2mul((3, 0), _f2mul(P[0], P[0])), _f2inv(_f2mul((2, 0), P[1])))
    else:
        lam = _f2mul(_f2sub(Q[1], P[1]), _f2inv(_f2sub(Q[0], P[0])))
    x3 = _f2sub(_f2sub(_f2mul(lam, lam), P[0]), Q[0])
    return (x3, _f2sub(_f2mul(lam, _f2sub(P[0], x3)), P[1]))


def g2_mul(k: int, P=G2_GENERATOR):
    k %= R_MOD
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, P)
    return acc


def g2_encode(P) -> np.ndarray:
    """affine G2 point (None = identity) -> (16,) uint64, the memory of `G2Affine`"""
    out = np.zeros(16, dtype=np.uint64)
    if P is not None:
        for c, v in enumerate((P[0][0], P[0][1], P[1][0], P[1][1])):
            m = v * MONT % Q_MOD
            out[4 * c:4 * c + 4] = [(m >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    return out


def g2_decode(arr: np.ndarray):
    a = np.asarray(arr, dtype=np.uint64).reshape(16)
    inv = pow(MONT, -1, Q_MOD)
    v = []
    for c in range(4):
        m = sum(int(a[4 * c + j]) << (64 * j) for j in range(4))
        if m >= Q_MOD:
            raise ValueError("G2 coordinate limbs are not a canonical Montgomery residue")
        v.append(m * inv % Q_MOD)
    if not any(v):
        return None
    return ((v[0], v[1]), (v[2], v[3]))


def _g1_sample_on_curve(points: np.ndarray, sample: int) -> None:
    """y^2 = x^3 + 3 on up to `sample` evenly spaced points (host big-int arithmetic; a whole-file check would be a GPU pass)"""
    n = points.shape[0]
    inv = pow(MONT, -1, Q_MOD)
    for i in sorted(set(np.linspace(0, n - 1, min(sample, n), dtype=np.int64).tolist())):
        w = [sum(int(points[i, 4 * c + j]) << (64 * j) for j in range(4)) for c in range(2)]
        if w[0] >= Q_MOD or w[1] >= Q_MOD:
            raise ValueError(f"point {i}: limbs are not a canonical Montgomery residue")
        x, y = w[0] * inv % Q_MOD, w[1] * inv % Q_MOD
        if (x, y) != (0, 0) and (y * y - x * x * x - 3) % Q_MOD:
            raise ValueError(f"point {i} is not on the curve")


def g1_first_invalid(points: np.ndarray) -> Optional[int]:
    """index of the first point of an (n, 8) G1Affine array that is not a valid curve point (non-canonical limbs or off the curve), None
    when all are valid: one GPU pass over the whole table (`zkhip_g1_check_points`), the check the reference's RawBytes reader runs"""
    import ctypes as C

    from . import _lib

    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    bad = C.c_uint64(0)
    _lib.check(_lib.load().zkhip_g1_check_points(pts.ctypes.data, pts.shape[0], C.byref(bad)))
    return None if bad.value >= pts.shape[0] else int(bad.value)


# ---- SerdeFormat::Processed: compressed points ------------------------------------------------------------------------------------
def g1_compress(points: np.ndarray, flag_layout: int = 0) -> bytes:
    """(n, 8) G1Affine array -> n x 32 bytes (`G1Affine::to_bytes` per point, on the GPU)"""
    from . import _lib

    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros(pts.shape[0] * 32, dtype=np.uint8)
    _lib.check(_lib.load().zkhip_g1_compress(pts.ctypes.data, pts.shape[0], out.ctypes.data, flag_layout))
    return out.tobytes()


def g1_decompress(data: bytes, n: int, flag_layout: int = 0, what: str = "point table") -> np.ndarray:
    """n x 32 bytes -> (n, 8) G1Affine array (`G1Affine::from_bytes` per point, on the GPU); raises on the first encoding that is not
    canonical or not on the curve"""
    import ctypes as C

    from . import _lib

    if len(data) != n * 32:
        raise ValueError(f"{what}: {len(data)} bytes where {n * 32} were expected")
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros((n, 8), dtype=np.uint64)
    bad = C.c_uint64(0)
    _lib.check(_lib.load().zkhip_g1_decompress(src.ctypes.data, n, out.ctypes.data, flag_layout, C.byref(bad)))
    if bad.value < n:
        raise ValueError(f"{what}: point {bad.value} does not decode to a curve point")
    return out


def _fq_sqrt(a: int) -> Optional[int]:
    r = pow(a, (Q_MOD + 1) // 4, Q_MOD)                      # q = 3 mod 4
    return r if r * r % Q_MOD == a % Q_MOD else None


def _f2sqrt(a: Fq2) -> Optional[Fq2]:
    """a root of a in Fq2 = Fq[u] / (u^2 + 1), None when a is not a square: with s^2 = a0^2 + a1^2, x0^2 = (a0 +- s) / 2 and
    x1 = a1 / (2 x0)"""
    a0, a1 = a[0] % Q_MOD, a[1] % Q_MOD
    if a1 == 0:
        r = _fq_sqrt(a0)
        if r is not None:
            return (r, 0)
        r = _fq_sqrt(-a0 % Q_MOD)                            # a0 = -(r^2) = (r u)^2
        return None if r is None else (0, r)
    s = _fq_sqrt((a0 * a0 + a1 * a1) % Q_MOD)
    if s is None:
        return None
    half = pow(2, -1, Q_MOD)
    for t in ((a0 + s) * half % Q_MOD, (a0 - s) * half % Q_MOD):
        x0 = _fq_sqrt(t)
        if x0:
            x = (x0, a1 * pow(2 * x0, -1, Q_MOD) % Q_MOD)
            if _f2mul(x, x) == (a0, a1):
                return x
    return None


def g2_compress(P, flag_layout: int = 0) -> bytes:
    """affine G2 point (None = identity) -> 64 bytes: x.c0 || x.c1 canonical little-endian, flags in the last byte; the sign is the lsb of
    the first byte of y's encoding, i.e. of y.c0"""
    if P is None:
        out = bytearray(64)
        if flag_layout == 0:
            out[63] |= 0x80
        return bytes(out)
    out = bytearray(P[0][0].to_bytes(32, "little") + P[0][1].to_bytes(32, "little"))
    out[63] |= (P[1][0] & 1) << (6 if flag_layout == 0 else 7)
    return bytes(out)


def g2_decompress(data: bytes, flag_layout: int = 0):
    if len(data) != 64:
        raise ValueError("a compressed G2 point is 64 bytes")
    b = bytearray(data)
    if flag_layout == 0:
        is_inf, sign = bool(b[63] >> 7), (b[63] >> 6) & 1
        b[63] &= 0x3F
    else:
        sign = b[63] >> 7
        b[63] &= 0x7F
        is_inf = not sign and not any(b)
    x = (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little"))
    if is_inf:
        if any(b) or sign:
            raise ValueError("G2: identity flag on a non-zero encoding")
        return None
    if x[0] >= Q_MOD or x[1] >= Q_MOD:
        raise ValueError("G2: x is not canonical")
    x3 = _f2mul(_f2mul(x, x), x)
    y = _f2sqrt(((x3[0] + G2_B[0]) % Q_MOD, (x3[1] + G2_B[1]) % Q_MOD))
    if y is None:
        raise ValueError("G2: x is not the abscissa of a point of the twist")
    if (y[0] & 1) != sign:
        y = (-y[0] % Q_MOD, -y[1] % Q_MOD)
    return (x, y)


RAW_BYTES, RAW_BYTES_UNCHECKED, PROCESSED = "RawBytes", "RawBytesUnchecked", "Processed"      # `SerdeFormat`


def write_params(f: BinaryIO, k: int, g: np.ndarray, g_lagrange: np.ndarray, g2: np.ndarray, s_g2: np.ndarray, fmt: str = RAW_BYTES,
                 flag_layout: int = 0) -> None:
    """`ParamsKZG::write_custom(writer, format)`; `write` is the RawBytes case"""
    n = 1 << k
    g = np.ascontiguousarray(g, dtype="<u8").reshape(-1, 8)
    g_lagrange = np.ascontiguousarray(g_lagrange, dtype="<u8").reshape(-1, 8)
    if g.shape[0] != n or g_lagrange.shape[0] != n:
        raise ValueError(f"g / g_lagrange must hold 2^{k} points")
    f.write(struct.pack("<I", k))
    if fmt == PROCESSED:
        f.write(g1_compress(g, flag_layout))
        f.write(g1_compress(g_lagrange, flag_layout))
        f.write(g2_compress(g2_decode(g2), flag_layout))
        f.write(g2_compress(g2_decode(s_g2), flag_layout))
        return
    if fmt not in (RAW_BYTES, RAW_BYTES_UNCHECKED):
        raise ValueError(f"unknown SerdeFormat {fmt!r}")
    f.write(g.tobytes())
    f.write(g_lagrange.tobytes())
    f.write(np.ascontiguousarray(g2, dtype="<u8").reshape(16).tobytes())
    f.write(np.ascontiguousarray(s_g2, dtype="<u8").reshape(16).tobytes())


def _read_params_processed(f: BinaryIO, k: int, flag_layout: int):
    n = 1 << k
    tabs = []
    for name in ("g", "g_lagrange"):
        buf = f.read(n * 32)
        if len(buf) != n * 32:
            raise ValueError(f"truncated SRS file: {name} holds {len(buf) // 32} of {n} points")
        tabs.append(g1_decompress(buf, n, flag_layout, name))
    tail = f.read(128)
    if len(tail) != 128:
        raise ValueError("truncated SRS file: g2 / s_g2 missing")
    return k, tabs[0], tabs[1], g2_encode(g2_decompress(tail[:64], flag_layout)), g2_encode(g2_decompress(tail[64:], flag_layout))


def read_params(f: BinaryIO, check_points: Optional[int] = None, max_k: int = 28, fmt: str = RAW_BYTES, flag_layout: int = 0):
    """-> (k, g, g_lagrange, g2, s_g2): `ParamsKZG::read_custom(reader, format)`.  RawBytes: `check_points` None (default) verifies EVERY
    point of both tables on the GPU and both G2 points on the host, like the reference's SerdeFormat::RawBytes reader; a positive number
    samples that many evenly spaced points on the host (no GPU needed); 0 checks nothing (RawBytesUnchecked, which is also what
    fmt = RAW_BYTES_UNCHECKED selects).  Processed: every point is decompressed (which is its validity check)."""
    head = f.read(4)
    if len(head) != 4:
        raise ValueError("truncated SRS file: no header")
    (k,) = struct.unpack("<I", head)
    if k > max_k:
        raise ValueError(f"SRS header says k = {k} (> {max_k}): not a KZG parameter file")
    if fmt == PROCESSED:
        return _read_params_processed(f, k, flag_layout)
    if fmt == RAW_BYTES_UNCHECKED:
        check_points = 0
    elif fmt != RAW_BYTES:
        raise ValueError(f"unknown SerdeFormat {fmt!r}")
    n = 1 << k

    def table(name: str) -> np.ndarray:
        buf = f.read(n * 64)
        if len(buf) != n * 64:
            raise ValueError(f"truncated SRS file: {name} holds {len(buf) // 64} of {n} points")
        return np.frombuffer(buf, dtype="<u8").reshape(n, 8).astype(np.uint64)

    g, g_lagrange = table("g"), table("g_lagrange")
    tail = f.read(256)
    if len(tail) != 256:
        raise ValueError("truncated SRS file: g2 / s_g2 missing")
    g2 = np.frombuffer(tail[:128], dtype="<u8").astype(np.uint64)
    s_g2 = np.frombuffer(tail[128:], dtype="<u8").astype(np.uint64)
    if check_points is None:
        for name, tab in (("g", g), ("g_lagrange", g_lagrange)):
            bad = g1_first_invalid(tab)
            if bad is not None:
                raise ValueError(f"{name}: point {bad} is not a valid curve point")
    elif check_points:
        _g1_sample_on_curve(g, check_points)
        _g1_sample_on_curve(g_lagrange, check_points)
    if check_points is None or check_points:
        for name, p in (("g2", g2), ("s_g2", s_g2)):
            if not g2_is_on_curve(g2_decode(p)):
                raise ValueError(f"{name} is not on the twist curve")
    return k, g, g_lagrange, g2, s_g2


def params_to_bytes(k: int, g, g_lagrange, g2, s_g2, fmt: str = RAW_BYTES) -> bytes:
    buf = io.BytesIO()
    write_params(buf, k, g, g_lagrange, g2, s_g2, fmt)
    return buf.getvalue()
