import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
import zksnap_circuits_halo2_amd as Z
lib = _lib.load()
for k, B in ((22, 26), (22, 1), (20, 26)):
    n = 1 << k
    dom = Z.EvaluationDomain(4, k)
    x = torch.randint(0, 1 << 62, (B, n, 4), dtype=torch.int64, device="cuda"); x[:, :, 3] &= (1 << 61) - 1
    f = lambda: _lib.check(lib.zkhip_ifft_scaled_batch_device(x.data_ptr(), dom.omega_inv.ctypes.data, k, dom.ifft_divisor.ctypes.data, B, n, None))
    t = time.perf_counter(); f(); torch.cuda.synchronize(); first = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); steady = (time.perf_counter() - t) / 3 * 1e3
    ext = torch.empty((B, 4 * n, 4), dtype=torch.int64, device="cuda")
    g = lambda: _lib.check(lib.zkhip_coeff_to_extended_device(x.data_ptr(), n, k, ext.data_ptr(), 4 * n, k + 2, B, dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data, None))
    t = time.perf_counter(); g(); torch.cuda.synchronize(); first2 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    for _ in range(3): g()
    torch.cuda.synchronize(); steady2 = (time.perf_counter() - t) / 3 * 1e3
    print(f"k={k} batch={B}: iNTT first {first:.2f} ms steady {steady:.2f} ms ({steady/B:.3f}/poly); coeff_to_extended first {first2:.2f} steady {steady2:.2f} ms ({steady2/B:.3f}/poly)", flush=True)
    del x, ext
