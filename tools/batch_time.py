import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(5)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for L, K in ((13, 256), (15, 64), (15, 412), (17, 16)):
    n = 1 << L
    a = rng.integers(0, 1 << 64, size=(K * n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=K * n, dtype=np.uint64)
    x = torch.from_numpy(a.view(np.int64)).cuda()
    om = F.fr_encode([F.omega_for(L)])[0]
    tb = timed(lambda: _lib.check(lib.zkhip_ntt_fr_batch_device(x.data_ptr(), om.ctypes.data, L, K, n, stream)))
    def loop():
        for b in range(K): _lib.check(lib.zkhip_ntt_fr_device(x.data_ptr() + b * n * 32, om.ctypes.data, L, stream))
    tl = timed(loop, 3)
    print(f"NTT 2^{L} x {K}: batched {tb:.3f} ms ({K*n/tb/1e3:.0f} Melem/s)   one call per polynomial {tl:.3f} ms ({K*n/tl/1e3:.0f} Melem/s)", flush=True)

t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L, K in ((13, 256), (15, 64), (15, 412), (17, 16), (20, 5)):
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), stream))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    a = rng.integers(0, 1 << 64, size=(K * n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=K * n, dtype=np.uint64)
    x = torch.from_numpy(a.view(np.int64)).cuda()
    out = torch.zeros(K * 12, dtype=torch.int64, device="cuda")
    tb = timed(lambda: _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, x.data_ptr(), n, K, n, out.data_ptr(), stream)), 5)
    def loop():
        for b in range(K): _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr() + b * n * 32, n, out.data_ptr() + b * 96, stream))
    tl = timed(loop, 2)
    print(f"MSM 2^{L} x {K}: batched {tb:.3f} ms ({K*n/tb/1e3:.0f} Mpoints/s)   one call per column {tl:.3f} ms ({K*n/tl/1e3:.0f} Mpoints/s)", flush=True)
    lib.zkhip_release_bases(h)
