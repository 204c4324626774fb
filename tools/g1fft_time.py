#!/usr/bin/env python3
"""Time g_to_lagrange (inverse FFT over G1 points) on device-resident points: tools/g1fft_time.py [k ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
for k in [int(x) for x in sys.argv[1:]] or [14, 16, 18, 20]:
    n = 1 << k
    g = torch.empty(n * 8, dtype=torch.int64, device="cuda"); out = torch.empty_like(g)
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, g.data_ptr(), None))
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        _lib.check(lib.zkhip_g_to_lagrange_device(g.data_ptr(), k, out.data_ptr(), None))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
    print(f"g_to_lagrange k={k}: {ms:9.1f} ms  ({n * k / 2 / ms / 1e3:.2f} M butterflies/s)", flush=True)
