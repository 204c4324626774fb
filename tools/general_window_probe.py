import os, sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L in (14, 15, 16, 17, 18, 19, 20):
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    line = f"2^{L} auto c={lib.zkhip_msm_window_bits(n)}:"
    for c in (11, 12, 13, 14, 15, 16):
        f = lambda: _lib.check(lib.zkhip_msm_g1_device_c(x.data_ptr(), bases.data_ptr(), n, out.data_ptr(), c, None))
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); line += f"  c={c}: {(time.perf_counter()-t)/5*1e3:.3f}"
    print(line, flush=True)
