#!/usr/bin/env python3
"""Host-buffer MSM against a registered SRS (the call an unmodified `create_proof` makes per commitment, /root/reference/aggregator/src/wrapper.rs:129),
PCIe-inclusive: median / min of `reps` calls at each size.   host_msm_ab.py [logs=21,22,23,24] [reps=15]
A/B of the chunked upload: run once plain and once under ZKHIP_STREAM_PIECE_LOG=0 (one upload, then the kernels) in the same gpurun call."""
import os, sys, time, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "21,22,23,24").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rng = np.random.default_rng(9)
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
tag = "one upload" if os.environ.get("ZKHIP_STREAM_PIECE_LOG") == "0" else "chunked (piece log %s)" % os.environ.get("ZKHIP_STREAM_PIECE_LOG", "20")
for L in logs:
    n = 1 << L
    d = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, d.data_ptr(), None)); torch.cuda.synchronize()
    bases = d.cpu().numpy().view(np.uint64).reshape(n, 8).copy(); del d
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.zkhip_register_bases(bases.ctypes.data, n))
    run = lambda: _lib.check(lib.zkhip_msm_g1(sc.ctypes.data, bases.ctypes.data, n, out.ctypes.data))
    for _ in range(3): run()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); run(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"host-buffer MSM 2^{L} registered, {tag}: median {st.median(ts):.3f} ms min {min(ts):.3f} ({n / st.median(ts) / 1e3:.0f} Mpoints/s) result {F.g1_decode_jacobian(out)[0] % 1000003}", flush=True)
    _lib.check(lib.zkhip_unregister_bases(bases.ctypes.data))
    del bases, sc
