#!/usr/bin/env python3
"""Phase times of the prepared MSM at explicit window widths: `scatter_probe.py LOG_N c1 c2 ...` (default 20; 10 12 13 14 15 16)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
def profile_read():
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    k = lib.zkhip_profile_read(ms, names, 32)
    return {names[i].value.decode(): ms[i] for i in range(max(k, 0))}
LOGN = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CS = [int(x) for x in sys.argv[2:]] or [10, 12, 13, 14, 15, 16]
n = 1 << LOGN
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), stream))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
sc = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
for c in CS:
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device_c(bases.data_ptr(), n, c, C.byref(h)))
    lib.zkhip_profile_enable(1)
    acc = {}
    for _ in range(3):
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
        for k, v in profile_read().items(): acc[k] = acc.get(k, 0) + v / 3
    lib.zkhip_profile_enable(0)
    W = (256 + c - 1) // c
    ent = W * n * (1 - 2.0 ** -c)
    print("   " + "  ".join(f"{k} {v*1e3:.0f}" for k, v in acc.items()) + "  (us)")
    print(f"c={c:2d} W={W:2d} LDS hist {4 << (min(c, 16) - 1) >> 10:4d} KiB  entries {ent/1e6:5.1f}M  count {acc['count']*1e3:7.1f} us ({acc['count']*1e9/ent:5.1f} ps/entry)  scatter {acc['scatter']*1e3:7.1f} us ({acc['scatter']*1e9/ent:5.1f} ps/entry)  accumulate {acc['accumulate']:.3f} ms total {sum(acc.values()):.3f}", flush=True)
    lib.zkhip_release_bases(h)
