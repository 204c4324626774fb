#!/usr/bin/env python3
"""Is a small prepared MSM bound by the host's enqueue rate?  Host time to enqueue one call vs the call's wall time."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L in (10, 13, 15, 17, 20):
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), stream))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    call = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), stream))
    call(); torch.cuda.synchronize()
    reps = 50
    t = time.perf_counter()
    for _ in range(reps): call()
    t_enq = (time.perf_counter() - t) / reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t) / reps
    # one call at a time (latency of a single MSM)
    t = time.perf_counter()
    for _ in range(reps): call(); torch.cuda.synchronize()
    t_one = (time.perf_counter() - t) / reps
    print(f"2^{L}: enqueue {t_enq*1e6:7.1f} us/call   back-to-back {t_all*1e6:7.1f} us/call   one at a time {t_one*1e6:7.1f} us/call", flush=True)
    lib.zkhip_release_bases(h)
