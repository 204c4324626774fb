#!/usr/bin/env python3
"""Large batched prepared MSMs (columns committed together, e.g. the advice columns of the wrapper circuit): K x 2^L."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L, Ks in ((20, (1, 2, 4, 8)), (22, (1, 2, 5))):
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), stream))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    for K in Ks:
        x = torch.randint(0, 1 << 62, (K * n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
        out = torch.zeros(K * 12, dtype=torch.int64, device="cuda")
        tb = timed(lambda: _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, x.data_ptr(), n, K, n, out.data_ptr(), stream)))
        print(f"MSM 2^{L} x {K}: batched {tb:.3f} ms = {tb/K:.3f} ms per MSM ({K*n/tb/1e3:.0f} Mpoints/s)", flush=True)
        del x
    lib.zkhip_release_bases(h)
    del bases
