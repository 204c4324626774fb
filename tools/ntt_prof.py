import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def profile_read():
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    return {names[i].value.decode(): round(ms[i], 4) for i in range(max(k, 0))}
rng = np.random.default_rng(1)
for L in [int(x) for x in (sys.argv[1:] or ["13", "15", "17", "20", "22", "24", "26"])]:
    N = 1 << L
    b = rng.integers(0, 1 << 64, size=(N, 4), dtype=np.uint64); b[:, 3] = rng.integers(0, 0x30644E72E131A029, size=N, dtype=np.uint64)
    x = torch.from_numpy(b.view(np.int64)).cuda()
    om = F.fr_encode([F.omega_for(L)])[0]
    _lib.check(lib.zkhip_ntt_fr_device(x.data_ptr(), om.ctypes.data, L, stream)); torch.cuda.synchronize()
    lib.zkhip_profile_enable(1)
    acc = {}
    for _ in range(5):
        _lib.check(lib.zkhip_ntt_fr_device(x.data_ptr(), om.ctypes.data, L, stream))
        for k, v in profile_read().items(): acc[k] = acc.get(k, 0) + v / 5
    lib.zkhip_profile_enable(0)
    tot = sum(acc.values())
    print(f"NTT 2^{L}: total {tot:.4f} ms  {N/tot/1e3:.0f} Melem/s  hbm_frac {64*N/(tot*1e-3)/8e12:.4f}", {k: round(v, 4) for k, v in acc.items()}, flush=True)
    del x
