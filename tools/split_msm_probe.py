#!/usr/bin/env python3
"""Would ONE large MSM gain from running as two point-range halves on two streams (each half a full prepared MSM against the same table at an
offset, one stream of high priority; the two 96-byte results added at the end)?  The second half's sort then runs under the first half's
accumulation and the first half's reduction tail under the second half's accumulation -- against that, two bucket reductions instead of one
and two accumulations of half the task count.  split_msm_probe.py [L ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
for L in [int(x) for x in (sys.argv[1:] or ["20", "21", "22", "24"])]:
    n = 1 << L
    t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1
    whole = torch.zeros(12, dtype=torch.int64, device="cuda")
    parts = torch.zeros(24, dtype=torch.int64, device="cuda")
    both = torch.zeros(12, dtype=torch.int64, device="cuda")
    main, side = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    half = n // 2

    def one():
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, whole.data_ptr(), C.c_void_p(main.cuda_stream)))

    def two():
        side.wait_stream(main)
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), half, parts.data_ptr(), C.c_void_p(main.cuda_stream)))
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, half, x.data_ptr() + half * 32, n - half, parts.data_ptr() + 96, C.c_void_p(side.cuda_stream)))
        main.wait_stream(side)
        _lib.check(lib.zkhip_g1_sum_device(parts.data_ptr(), 2, both.data_ptr(), C.c_void_p(main.cuda_stream)))

    res = {}
    for rep in range(2):
        for name, f in (("one", one), ("two halves", two)):
            for _ in range(3): f()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(12): f()
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t) * 1e3 / 12)
    same = F.g1_decode_jacobian(whole.cpu().numpy().view(np.uint64)[:12]) == F.g1_decode_jacobian(both.cpu().numpy().view(np.uint64)[:12])
    print(f"2^{L}: one launch set " + " / ".join(f"{v:.3f}" for v in res["one"]) + " ms;  two halves on two streams " + " / ".join(f"{v:.3f}" for v in res["two halves"]) + f" ms;  results equal: {same}", flush=True)
    lib.zkhip_release_bases(h)
    del bases, x
