"""Host-side constants and (de)serialisation of the boundary formats -- the part of halo2curves'
`bn256::{Fr,Fq,G1Affine,G1}` [DEP] a host needs in order to *call* the kernels: Montgomery limbs in and out,
roots of unity, domain constants.  Scalar big-int arithmetic only; no vector arithmetic happens here."""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B97091_43E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D_3C208C16D87CFD47
MONT = 1 << 256
S = 28
ROOT_OF_UNITY = pow(7, (R_MOD - 1) >> S, R_MOD)
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
_MASK = (1 << 64) - 1


def _limbs(v: int):
    return [(v >> (64 * i)) & _MASK for i in range(4)]


def fr_encode(values: Iterable[int]) -> np.ndarray:
    """ints -> (n, 4) uint64 Montgomery limbs (the memory of `&[Fr]`)."""
    vals = [_limbs((int(v) % R_MOD) * MONT % R_MOD) for v in values]
    return np.array(vals, dtype=np.uint64).reshape(len(vals), 4)


def fr_decode(arr: np.ndarray):
    inv = pow(MONT, -1, R_MOD)
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, j]) << (64 * j) for j in range(4)) * inv % R_MOD for i in range(a.shape[0])]


def fr_one_limbs() -> np.ndarray:
    return fr_encode([1])[0]


def g1_encode(points: Sequence[Optional[Tuple[int, int]]]) -> np.ndarray:
    """affine points (None = identity) -> (n, 8) uint64 (the memory of `&[G1Affine]`)."""
    out = np.zeros((len(points), 8), dtype=np.uint64)
    for i, P in enumerate(points):
        if P is None:
            continue
        out[i, 0:4] = _limbs(P[0] * MONT % Q_MOD)
        out[i, 4:8] = _limbs(P[1] * MONT % Q_MOD)
    return out


def g1_decode_jacobian(xyz: np.ndarray) -> Optional[Tuple[int, int]]:
    """(12,) uint64 Jacobian Montgomery limbs (the memory of `G1`) -> affine point or None."""
    inv = pow(MONT, -1, Q_MOD)
    a = np.asarray(xyz, dtype=np.uint64).reshape(12)
    X, Y, Z = (sum(int(a[4 * c + j]) << (64 * j) for j in range(4)) for c in range(3))
    assert X < Q_MOD and Y < Q_MOD and Z < Q_MOD, "non-canonical limbs"
    X, Y, Z = X * inv % Q_MOD, Y * inv % Q_MOD, Z * inv % Q_MOD
    if Z == 0:
        return None
    zi = pow(Z, -1, Q_MOD)
    return (X * zi * zi % Q_MOD, Y * zi * zi * zi % Q_MOD)


def omega_for(log_n: int) -> int:
    assert 0 <= log_n <= S
    return pow(ROOT_OF_UNITY, 1 << (S - log_n), R_MOD)
