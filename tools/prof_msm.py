#!/usr/bin/env python3
"""Workload for rocprofv3: a few back-to-back prepared MSMs at 2^20 (the bench configuration) and one NTT 2^24."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << L
stream = torch.cuda.current_stream().cuda_stream
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), stream))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
sc = torch.from_numpy(a.view(np.int64)).cuda()
out = torch.zeros(12, dtype=torch.int64, device="cuda")
h = C.c_uint64(0)
_lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
for _ in range(reps):
    _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
torch.cuda.synchronize()
if len(sys.argv) > 3:
    LN = int(sys.argv[3]); N = 1 << LN
    b = rng.integers(0, 1 << 64, size=(N, 4), dtype=np.uint64); b[:, 3] = rng.integers(0, 0x30644E72E131A029, size=N, dtype=np.uint64)
    x = torch.from_numpy(b.view(np.int64)).cuda()
    om = F.fr_encode([F.omega_for(LN)])[0]
    for _ in range(reps):
        _lib.check(lib.zkhip_ntt_fr_device(x.data_ptr(), om.ctypes.data, LN, stream))
    torch.cuda.synchronize()
print("done")
