#!/usr/bin/env python3
"""Prepared MSM of 2^L points (small, latency-bound) under every window size: which c is fastest end to end?  small_window_probe.py [L ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
for L in [int(x) for x in (sys.argv[1:] or ["13", "15", "17"])]:
    n = 1 << L
    t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    line = []
    for c in [0] + list(range(7, 17)):
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device_c(bases.data_ptr(), n, c, C.byref(h)))
        f = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), None))
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 30 * 1e3
        line.append(f"c={lib.zkhip_prepared_window_bits(h) if c == 0 else c}{'(auto)' if c == 0 else ''} {ms:.3f}")
        lib.zkhip_release_bases(h)
    print(f"2^{L}: " + "  ".join(line), flush=True)
