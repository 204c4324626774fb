"""ctypes binding of oracle/libcpu_ref.so (the C restatement in cpu_ref.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcpu_ref.so")
_lib = None


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "libcpu_ref.so"], stdout=subprocess.DEVNULL)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def field_op(field: int, op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros_like(a)
    load().ref_field_op(field, op, _p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def best_multiexp(scalars: np.ndarray, bases: np.ndarray, threads: int = 1) -> np.ndarray:
    n = scalars.shape[0]
    assert bases.shape[0] == n
    out = np.zeros(12, dtype=np.uint64)
    load().ref_best_multiexp(_p(scalars), _p(bases), C.c_size_t(n), int(threads), _p(out))
    return out


def best_fft(a: np.ndarray, omega: np.ndarray, log_n: int, threads: int = 1) -> None:
    assert a.shape[0] == 1 << log_n and a.flags.c_contiguous
    load().ref_best_fft(_p(a), _p(omega), C.c_uint32(log_n), int(threads))


def scale(a: np.ndarray, c: np.ndarray) -> None:
    load().ref_scale(_p(a), C.c_size_t(a.shape[0]), _p(c))


def distribute_powers_zeta(a: np.ndarray, p1: np.ndarray, p2: np.ndarray) -> None:
    load().ref_distribute_powers_zeta(_p(a), C.c_size_t(a.shape[0]), _p(p1), _p(p2))


def mul_periodic(a: np.ndarray, table: np.ndarray) -> None:
    load().ref_mul_periodic(_p(a), C.c_size_t(a.shape[0]), _p(table), C.c_size_t(table.shape[0]))


def eval_polynomial(poly: np.ndarray, point: np.ndarray) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    load().ref_eval_polynomial(_p(poly), C.c_size_t(poly.shape[0]), _p(point), _p(out))
    return out


def eval_polynomial_mt(poly: np.ndarray, point: np.ndarray, threads: int) -> np.ndarray:
    """halo2's multi-core `eval_polynomial` (chunk per thread, parts scaled by point^start and summed)"""
    out = np.zeros(4, dtype=np.uint64)
    load().ref_eval_polynomial_mt(_p(poly), C.c_size_t(poly.shape[0]), _p(point), int(threads), _p(out))
    return out


def usable_cores() -> int:
    """host threads this process may actually run at once: the scheduler affinity mask capped by the cgroup CPU quota
    (what rayon's default pool size -- std::thread::available_parallelism -- reports on the same box)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:   # noqa: BLE001
            continue
    return max(1, n)


def kate_division(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    q = np.zeros((max(a.shape[0] - 1, 0), 4), dtype=np.uint64)
    load().ref_kate_division(_p(a), C.c_size_t(a.shape[0]), _p(b), _p(q))
    return q


def batch_invert(a: np.ndarray) -> None:
    load().ref_batch_invert(_p(a), C.c_size_t(a.shape[0]))


def prefix_product(v: np.ndarray) -> np.ndarray:
    out = np.zeros_like(v)
    load().ref_prefix_product(_p(v), C.c_size_t(v.shape[0]), _p(out))
    return out


def jac_to_affine(xyz: np.ndarray) -> np.ndarray:
    out = np.zeros(8, dtype=np.uint64)
    load().ref_jac_to_affine(_p(np.ascontiguousarray(xyz, dtype=np.uint64)), _p(out))
    return out


def jac_add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    load().ref_jac_add(_p(a), _p(b), _p(out))
    return out


def scalar_mul(k_canon: int, base: np.ndarray) -> np.ndarray:
    k = np.array([(k_canon >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    load().ref_scalar_mul(_p(k), _p(np.ascontiguousarray(base, dtype=np.uint64)), _p(out))
    return out


def gen_scalars(seed: int, n: int, kind: int = 0) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    load().ref_gen_scalars(C.c_uint64(seed), C.c_size_t(n), int(kind), _p(out))
    return out


def gen_bases(seed: int, n: int):
    """bases[i] = (t0 + i*d) * G.  Returns (bases (n,8) uint64, t0, d) with t0, d canonical ints."""
    out = np.zeros((n, 8), dtype=np.uint64)
    t0 = np.zeros(4, dtype=np.uint64)
    d = np.zeros(4, dtype=np.uint64)
    load().ref_gen_bases(C.c_uint64(seed), C.c_size_t(n), _p(out), _p(t0), _p(d))
    toint = lambda v: sum(int(v[i]) << (64 * i) for i in range(4))
    return out, toint(t0), toint(d)


def expected_scalar(scalars: np.ndarray, t0: int, d: int) -> int:
    """sum_i scalars[i] * (t0 + i d) mod r  -- MSM(scalars, gen_bases) must equal this multiple of G."""
    lim = lambda v: np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    load().ref_expected_scalar(_p(scalars), C.c_size_t(scalars.shape[0]), _p(lim(t0)), _p(lim(d)), _p(out))
    return sum(int(out[i]) << (64 * i) for i in range(4))


GENERATOR = None


def generator() -> np.ndarray:
    """G = (1, 2) in G1Affine memory format."""
    global GENERATOR
    if GENERATOR is None:
        q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
        m = lambda v: [((v << 256) % q >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]
        GENERATOR = np.array(m(1) + m(2), dtype=np.uint64)
    return GENERATOR
