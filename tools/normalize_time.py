#!/usr/bin/env python3
"""median time of zkhip_g1_batch_normalize_device for a few batch sizes: normalize_time.py [label]"""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
t0, dd = F.fr_encode([77])[0], F.fr_encode([991])[0]
for n in (16, 427, 4096, 1 << 16, 1 << 20):
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), None))
    jac = torch.zeros((n, 12), dtype=torch.int64, device="cuda")
    Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    one = np.array([((1 << 256) % Q >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)       # z = 1 (Montgomery-256)
    jac[:, :8] = bases.view(n, 8); jac[:, 8:12] = torch.from_numpy(one.view(np.int64)).cuda()
    out = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    def run(): _lib.check(lib.zkhip_g1_batch_normalize_device(jac.data_ptr(), n, out.data_ptr(), None))
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{sys.argv[1] if len(sys.argv) > 1 else '':6s} batch_normalize n={n}: median {st.median(ts):.4f} ms", flush=True)
