import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
for L in (13, 15, 17):
    n = 1 << L
    t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    f = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), None))
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    lib.zkhip_profile_enable(1); f()
    tm = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)(); k = lib.zkhip_profile_read(tm, names, 32); lib.zkhip_profile_enable(0)
    print(f"2^{L} c={lib.zkhip_prepared_window_bits(h)} {ms:.3f} ms  " + " ".join(f"{names[i].value.decode()}={tm[i]:.3f}" for i in range(k)), flush=True)
    lib.zkhip_release_bases(h)
