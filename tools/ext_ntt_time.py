#!/usr/bin/env python3
"""median time of coeff_to_extended (n = 2^k coefficients -> the 4n-point coset, one zero-padded transform): ext_ntt_time.py [k=22] [reps=30] [label]"""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F, domain as D
lib = _lib.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, ek = 1 << k, k + 2
en = 1 << ek
dom = D.EvaluationDomain(4, k)
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
x = torch.from_numpy(a.view(np.int64)).cuda()
ext = torch.empty((en, 4), dtype=torch.int64, device="cuda")
def run(): _lib.check(lib.zkhip_coeff_to_extended_device(x.data_ptr(), n, k, ext.data_ptr(), en, ek, 1, dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data, None))
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"{sys.argv[3] if len(sys.argv) > 3 else '':8s} coeff_to_extended 2^{k} -> 2^{ek}: median {st.median(ts):.4f} ms min {min(ts):.4f}")
