"""PCIe-inclusive timing of the host-buffer entry points (what the two-function Rust drop-in would see)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, ctypes as C
import zksnap_circuits_halo2_amd as Z
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
rng = np.random.default_rng(9)
def rand_fr(n):
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64); return a
def timed(fn, reps=5):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e3
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L in (15, 20, 22):
    n = 1 << L
    d = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, d.data_ptr(), None)); torch.cuda.synchronize()
    bases = d.cpu().numpy().view(np.uint64).reshape(n, 8).copy(); del d
    sc = rand_fr(n)
    params = Z.ParamsKZG(L, bases)
    t_reg = timed(lambda: params.commit(sc))
    params.close()
    t_unreg = timed(lambda: Z.best_multiexp(sc, bases), 2)
    print(f"host API MSM 2^{L}: registered SRS {t_reg:.2f} ms ({n/t_reg/1e3:.0f} Mpoints/s), unregistered bases {t_unreg:.2f} ms", flush=True)
for L in (15, 20, 22, 24):
    n = 1 << L
    a = rand_fr(n); om = F.fr_encode([F.omega_for(L)])[0]
    t = timed(lambda: Z.best_fft(a, om, L), 3)
    print(f"host API NTT 2^{L}: {t:.2f} ms ({n/t/1e3:.0f} Melem/s, {64*n/(t*1e-3)/1e9:.1f} GB/s over PCIe both ways)", flush=True)
