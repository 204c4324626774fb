"""GPU: the bucket-free "direct table" MSM of small fixed base sets (round 4; csrc/msm.hip "Direct tables").

Prepared / registered sets of at most 2^15 points (the voter and state-transition circuits' SRS: k = 13 / 15,
/root/reference/voter/benches/voter_circuit.rs:23, /root/reference/aggregator/benches/state_transition_circuit.rs:22) get a table of all
128 multiples of every 8-bit window point, and a single MSM over them is a plain sum of 32 n table points: 8-bit signed digits recoded inside the
accumulation kernel, 4 gathers + 3 mixed additions per thread, a quad-cooperative tree.  Same group element as `best_multiexp`: checked against the C
oracle's restatement, the structured-SRS identity and this library's own bucket path (explicit window), with the scalars that stress the recoding."""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F

pytestmark = pytest.mark.gpu
R = O.R_MOD
D = 0x9E3779B97F4A7C15F39CC0605CEDC835
T0 = 0x5A4B534E41500002 + 5005


def walk(lib, n, torch):
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    torch.cuda.synchronize()
    return bases


def expect(cref, sc, t0=T0):
    return cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(np.ascontiguousarray(sc), t0, D), cref.generator()))


def phases(lib):
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    return [names[i].value.decode() for i in range(max(k, 0))]


def edge_scalars(rng, n):
    """values whose bytes sit on the recoding's edges: 0x7f / 0x80 / 0xff runs (carries that ripple through every window), powers of two, r - 1 ..."""
    vals = [0, 1, 2, 127, 128, 129, 255, 256, 0x7F7F7F7F, 0x80808080, R - 1, R - 2, R - 128, R - 129, (R - 1) // 2, (R + 1) // 2,
            int.from_bytes(b"\x7f" * 31, "little"), int.from_bytes(b"\x80" * 31, "little"), int.from_bytes(b"\xff" * 31, "little"),
            int.from_bytes(b"\xff" * 16 + b"\x7f" * 15, "little"), (1 << 253) - 1, (1 << 253), (1 << 253) + (1 << 252)]
    vals += [1 << k for k in range(0, 254, 7)] + [(1 << k) - 1 for k in range(1, 254, 9)] + [(0x80 << (8 * k)) % R for k in range(31)]
    vals += [rng.randrange(R) for _ in range(max(0, n - len(vals)))]
    rng.shuffle(vals)
    return F.fr_encode([v % R for v in vals[:n]])


@pytest.mark.parametrize("n", [2, 3, 17, 100, 1000, 1025, 2049, 4097, 1 << 13, (1 << 13) + 5, (1 << 14) + 1, 1 << 15])
def test_direct_msm_vs_oracle_and_bucket_path(lib, cref, n):
    import random

    import torch

    rng = random.Random(600 + n)
    d_bases = walk(lib, n, torch)
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(d_bases.data_ptr(), n, C.byref(h)))          # automatic window: gets the direct table
    hb = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device_c(d_bases.data_ptr(), n, 13 if n > 64 else 8, C.byref(hb)))   # explicit window: buckets only
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    host_bases = d_bases.cpu().numpy().view(np.uint64).reshape(n, 8)
    try:
        for kind in ("edge", "uniform", "ones_and_zeros"):
            if kind == "edge":
                sc = edge_scalars(rng, n)
            elif kind == "uniform":
                sc = cref.gen_scalars(601 + n, n, 0)
            else:
                sc = np.zeros((n, 4), dtype=np.uint64); sc[::3] = F.fr_encode([1])[0]; sc[1::3] = F.fr_encode([R - 1])[0]
            d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
            lib.zkhip_profile_enable(1)
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, d_sc.data_ptr(), n, out.data_ptr(), stream))
            torch.cuda.synchronize()
            assert "direct_accumulate" in phases(lib), "the automatic table of a small set must take the direct path"
            lib.zkhip_profile_enable(0)
            got = cref.jac_to_affine(out.cpu().numpy().view(np.uint64))
            assert np.array_equal(got, expect(cref, sc)), (n, kind)
            _lib.check(lib.zkhip_msm_g1_prepared_device(hb, 0, d_sc.data_ptr(), n, out.data_ptr(), stream))
            torch.cuda.synchronize()
            assert np.array_equal(cref.jac_to_affine(out.cpu().numpy().view(np.uint64)), got), (n, kind, "bucket path")
            if n <= 1000:     # the reference's algorithm itself (chunked Pippenger, c = ceil(ln n)) on the same inputs
                assert np.array_equal(cref.jac_to_affine(cref.best_multiexp(np.ascontiguousarray(sc), host_bases, 2)), got), (n, kind, "oracle")
        # sub-ranges of the prepared set (ParamsKZG::commit of a shorter polynomial, offsets into a shard)
        sc = cref.gen_scalars(777 + n, n, 0)
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        for lo, m in ((0, n - 1), (1, n - 1), (n // 3, n - n // 3), (n - 1, 1), (0, 0)):
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, lo, d_sc.data_ptr(), m, out.data_ptr(), stream))
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint64)
            if m == 0:
                assert F.g1_decode_jacobian(got) is None
            else:
                assert np.array_equal(cref.jac_to_affine(got), expect(cref, sc[:m], (T0 + lo * D) % R)), (n, lo, m)
    finally:
        lib.zkhip_release_bases(h)
        lib.zkhip_release_bases(hb)


def test_direct_msm_with_identity_bases_and_repeated_points(lib, cref):
    """(0, 0) bases contribute nothing; equal points meet in the tree (the doubling branch of the general addition), opposite points cancel"""
    import torch

    n = 512
    d_bases = walk(lib, n, torch)
    hb = d_bases.cpu().numpy().view(np.uint64).reshape(n, 8).copy()
    hb[7] = 0; hb[300] = 0                               # identities
    hb[11] = hb[10]; hb[12] = hb[10]                      # repeated point
    q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    neg = hb[20].copy()
    y = sum(int(neg[4 + i]) << (64 * i) for i in range(4))
    ny = (q - y) % q
    neg[4:] = [(ny >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]
    hb[21] = neg                                          # hb[21] = -hb[20]
    sc = cref.gen_scalars(888, n, 0)
    sc[11] = sc[10]; sc[12] = sc[10]; sc[21] = sc[20]     # equal scalars on equal / opposite points: the same table entries meet
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.zkhip_register_bases(hb.ctypes.data, n))
    try:
        _lib.check(lib.zkhip_msm_g1(sc.ctypes.data, hb.ctypes.data, n, out.ctypes.data))
    finally:
        _lib.check(lib.zkhip_unregister_bases(hb.ctypes.data))
    exp = cref.jac_to_affine(cref.best_multiexp(sc, hb, 2))
    assert np.array_equal(cref.jac_to_affine(out), exp)


def test_direct_path_through_registered_host_arrays_and_batches(lib, cref):
    """zkhip_register_bases on a 2^13-point array: the single commit takes the direct table, the batched commit the shared-bucket launch set; both
    must give the structured identity"""
    import torch

    n = 1 << 13
    d_bases = walk(lib, n, torch)
    hb = np.ascontiguousarray(d_bases.cpu().numpy().view(np.uint64).reshape(n, 8))
    scs = np.ascontiguousarray(np.stack([cref.gen_scalars(900 + k, n, 0) for k in range(3)]))
    _lib.check(lib.zkhip_register_bases(hb.ctypes.data, n))
    try:
        single = np.zeros((3, 12), dtype=np.uint64)
        for k in range(3):
            _lib.check(lib.zkhip_msm_g1(scs[k].ctypes.data, hb.ctypes.data, n, single[k].ctypes.data))
        batch = np.zeros((3, 12), dtype=np.uint64)
        _lib.check(lib.zkhip_msm_g1_batch(scs.ctypes.data, hb.ctypes.data, n, 3, batch.ctypes.data))
        for k in range(3):
            e = expect(cref, scs[k])
            assert np.array_equal(cref.jac_to_affine(single[k]), e) and np.array_equal(cref.jac_to_affine(batch[k]), e)
    finally:
        _lib.check(lib.zkhip_unregister_bases(hb.ctypes.data))


def test_direct_tables_respect_the_budget():
    """(round 4 advice) a direct table is 256 KiB per point: it is built only within $ZKHIP_DIRECT_BUDGET_GIB (all tables of the process) and half of
    the free HBM.  With a budget of 1 GiB a 2^11-point set (512 MiB) gets its table, a second set of 2^12 points (1 GiB more) does not and keeps
    the bucket path -- with the same result.  The knob is read once per process: a child process."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np, torch
from oracle import cpu_ref as cref
from zksnap_circuits_halo2_amd import _lib, fields as F
cref.load()
lib = _lib.load()
D, T0 = 0x9E3779B97F4A7C15F39CC0605CEDC835, 0x5A4B534E41500002 + 4242
def run(n):
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    torch.cuda.synchronize()
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    sc = cref.gen_scalars(900 + n, n, 0)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    lib.zkhip_profile_enable(1)
    _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, d_sc.data_ptr(), n, out.data_ptr(), None))
    torch.cuda.synchronize()
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    lib.zkhip_profile_enable(0)
    ph = [names[i].value.decode() for i in range(k)]
    want = cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, T0, D), cref.generator()))
    assert np.array_equal(cref.jac_to_affine(out.cpu().numpy().view(np.uint64)), want), n
    return h, "direct_accumulate" in ph
h1, d1 = run(1 << 11)
h2, d2 = run(1 << 12)
lib.zkhip_release_bases(h1)
h3, d3 = run(1 << 11)            # the first table's bytes were returned to the budget
print("DIRECT", d1, d2, d3)
''' % root
    env = dict(os.environ, ZKHIP_DIRECT_BUDGET_GIB="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "DIRECT True False True" in res.stdout, res.stdout
