#!/usr/bin/env python3
"""Do independent large MSMs overlap when they are enqueued on two streams?  (One MSM's latency-bound reduction tail under the other's
accumulation.)  Prepared MSM 2^L, 8 MSMs: one stream vs two."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
for L in (20, 22):
    n = 1 << L
    t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    xs = []
    for i in range(2):
        x = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda"); x[:, 3] &= (1 << 61) - 1; xs.append(x)
    outs = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(2)]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    def run(streams, reps=8):
        for i in range(reps):
            st = streams[i % len(streams)]
            _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, xs[i % 2].data_ptr(), n, outs[i % 2].data_ptr(), C.c_void_p(st.cuda_stream)))
    class Default:
        cuda_stream = 0
    for name, streams in (("default stream", [Default]), ("one stream", [s0]), ("two streams", [s0, s1]), ("default stream", [Default]), ("one stream", [s0]),
                          ("two streams", [s0, s1]), ("default + one", [Default, s1])):
        run(streams, 18); torch.cuda.synchronize(); t = time.perf_counter()
        run(streams, 18); torch.cuda.synchronize()
        print(f"2^{L} 18 MSMs, {name}: {(time.perf_counter() - t) * 1e3 / 18:.3f} ms per MSM", flush=True)
    lib.zkhip_release_bases(h)
