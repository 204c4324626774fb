#!/usr/bin/env python3
"""median time + phases of the general-path MSM (unregistered, device-resident bases): general_path_time.py [log_n=20] [reps=40] [label]"""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = 1 << L
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), None))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
sc = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
def run(): _lib.check(lib.zkhip_msm_g1_device(sc.data_ptr(), bases.data_ptr(), n, out.data_ptr(), None))
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)(); acc = {}
lib.zkhip_profile_enable(1)
for _ in range(15):
    run(); torch.cuda.synchronize()
    k = lib.zkhip_profile_read(ms, names, 32)
    for i in range(k): acc.setdefault(names[i].value.decode(), []).append(ms[i])
lib.zkhip_profile_enable(0)
print(f"{sys.argv[3] if len(sys.argv) > 3 else '':8s} general 2^{L} median {st.median(ts):.4f} min {min(ts):.4f} | " + " ".join(f"{k} {st.median(v):.4f}" for k, v in acc.items()))
