import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for L in (13, 15, 20, 22, 24):
    n = 1 << L
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    x = torch.from_numpy(a.view(np.int64)).cuda(); y = torch.empty_like(x); r = torch.zeros(4, dtype=torch.int64, device="cuda")
    pt = F.fr_encode([0x1234567890ABCDEF1234567890ABCDEF])[0]
    te = timed(lambda: _lib.check(lib.zkhip_fr_eval_polynomial_device(x.data_ptr(), n, pt.ctypes.data, r.data_ptr(), stream)))
    tk = timed(lambda: _lib.check(lib.zkhip_fr_kate_division_device(x.data_ptr(), n, pt.ctypes.data, y.data_ptr(), stream)))
    tp = timed(lambda: _lib.check(lib.zkhip_fr_prefix_product_device(x.data_ptr(), n, y.data_ptr(), stream)))
    ti = timed(lambda: _lib.check(lib.zkhip_fr_batch_invert_device(y.data_ptr(), n, stream)))
    gb = lambda ms, bytes_per: bytes_per * n / (ms * 1e-3) / 1e9
    print(f"2^{L}: eval_polynomial {te:.3f} ms ({gb(te,32):.0f} GB/s alg)  kate_division {tk:.3f} ms ({gb(tk,64):.0f} GB/s)  prefix_product {tp:.3f} ms ({gb(tp,64):.0f} GB/s)  batch_invert {ti:.3f} ms ({gb(ti,64):.0f} GB/s)", flush=True)
