#!/usr/bin/env python3
"""Does a measurement depend on how long the GPU has been busy?  Per-call HIP-event times of back-to-back prepared MSMs (2^20) and NTTs (2^24)
after an idle second: ramp_probe.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load(); stream = torch.cuda.current_stream().cuda_stream
n = 1 << 20
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), stream))
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
sc = torch.from_numpy(a.view(np.int64)).cuda(); out = torch.zeros(12, dtype=torch.int64, device="cuda")
h = C.c_uint64(0); _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
N = 1 << 24
b = rng.integers(0, 1 << 64, size=(N, 4), dtype=np.uint64); b[:, 3] = rng.integers(0, 0x30644E72E131A029, size=N, dtype=np.uint64)
x = torch.from_numpy(b.view(np.int64)).cuda(); om = F.fr_encode([F.omega_for(24)])[0]
def msm(): _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
def ntt(): _lib.check(lib.zkhip_ntt_fr_device(x.data_ptr(), om.ctypes.data, 24, stream))
for name, fn, cnt in (("msm 2^20", msm, 120), ("ntt 2^24", ntt, 60)):
    fn(); torch.cuda.synchronize(); time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(cnt + 1)]
    ev[0].record()
    for i in range(cnt): fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(cnt)]
    grp = lambda lo, hi: sum(ts[lo:hi]) / (hi - lo)
    print(f"{name}: calls 0-4 {grp(0,5):.4f} ms, 5-9 {grp(5,10):.4f}, 10-19 {grp(10,20):.4f}, 20-39 {grp(20,40):.4f}, 40-{cnt-1} {grp(40,cnt):.4f}", flush=True)
