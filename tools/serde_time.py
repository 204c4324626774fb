#!/usr/bin/env python3
"""G1 point compression / decompression on the device (`SerdeFormat::Processed`): time per 2^L points, device-resident."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
for L in (16, 20, 22):
    n = 1 << L
    t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
    pts = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, pts.data_ptr(), None))
    comp = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    back = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    bad = C.c_uint64(0)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    tc = timed(lambda: _lib.check(lib.zkhip_g1_compress_device(pts.data_ptr(), n, comp.data_ptr(), 0, None)))
    td = timed(lambda: _lib.check(lib.zkhip_g1_decompress_device(comp.data_ptr(), n, back.data_ptr(), 0, C.byref(bad), None)))
    assert bad.value == n and torch.equal(pts, back)
    print(f"2^{L}: compress {tc:.3f} ms ({n / tc / 1e3:.0f} Mpoints/s)   decompress {td:.3f} ms ({n / td / 1e3:.1f} Mpoints/s)", flush=True)
