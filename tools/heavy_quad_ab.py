#!/usr/bin/env python3
"""Prepared MSM on the witness-like scalar mix (60 % zero, 30 % below 2^88, 10 % uniform: SURVEY.md section 8(d) config 2 (W)) and on a column of 88-bit
limbs only: median time and phases.  Run once plain and once under ZKHIP_HEAVY_QUAD_MAX=0 (the 128-lane slice sums for every heavy bucket) in one gpurun call.
  heavy_quad_ab.py [logs=20,22] [reps=60]"""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,22").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
R = F.R_MOD
stream = torch.cuda.current_stream().cuda_stream
tag = "quad_max=" + os.environ.get("ZKHIP_HEAVY_QUAD_MAX", "256(default)")
def mont(plain):
    r2 = np.ascontiguousarray(np.broadcast_to(F.fr_encode([pow(2, 256, R)])[0], plain.shape)); out = np.empty_like(plain)
    _lib.check(lib.zkhip_test_field_op(1, 0, plain.ctypes.data, r2.ctypes.data, out.ctypes.data, plain.shape[0])); return out
def mix(n, kind, rng):
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    sel = rng.integers(0, 10, size=n)
    small = np.flatnonzero((sel >= 6) & (sel < 9)) if kind == "witness" else np.arange(n)
    if kind == "witness": a[sel < 6] = 0
    pl = np.zeros((len(small), 4), dtype=np.uint64); pl[:, 0] = a[small, 0]; pl[:, 1] = a[small, 1] & np.uint64((1 << 24) - 1)
    a[small] = mont(pl)
    return a
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L in logs:
    n = 1 << L
    d = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, d.data_ptr(), stream)); torch.cuda.synchronize()
    h = C.c_uint64(0); _lib.check(lib.zkhip_prepare_bases_device(d.data_ptr(), n, C.byref(h)))
    for kind in ("witness", "limbs88"):
        sc = torch.from_numpy(mix(n, kind, np.random.default_rng(5 + L)).view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        run = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
        for _ in range(15): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)(); acc = {}
        lib.zkhip_profile_enable(1)
        for _ in range(12):
            run(); torch.cuda.synchronize(); k = lib.zkhip_profile_read(ms, names, 32)
            for i in range(k): acc.setdefault(names[i].value.decode(), []).append(ms[i])
        lib.zkhip_profile_enable(0)
        pt = F.g1_decode_jacobian(out.cpu().numpy().view(np.uint64))
        print(f"2^{L} {kind:8s} {tag}: median {st.median(ts):.4f} ms min {min(ts):.4f} | " + " ".join(f"{k} {st.median(v):.4f}" for k, v in acc.items()) + f" | x mod 1e6+3 = {pt[0] % 1000003}", flush=True)
    lib.zkhip_release_bases(h)
