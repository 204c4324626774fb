#!/usr/bin/env python3
"""Small prepared MSMs one at a time (what an unmodified create_proof issues per column at k = 13 .. 15): the bucket-free direct table (automatic for
n <= 2^15) against the bucket path (explicit window = the bucket path's own best choice), device-resident and through host buffers.
  small_msm_ab.py [logs=10,12,13,14,15] [reps=200]"""
import os, sys, time, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,12,13,14,15").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)
BUCKET_C = {10: 13, 11: 13, 12: 13, 13: 13, 14: 15, 15: 15, 16: 16}
def timed(run):
    for _ in range(20): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return st.median(ts), min(ts)
def phases(run):
    ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
    lib.zkhip_profile_enable(1)
    acc = {}
    for _ in range(20):
        run(); torch.cuda.synchronize()
        k = lib.zkhip_profile_read(ms, names, 32)
        for i in range(k): acc.setdefault(names[i].value.decode(), []).append(ms[i])
    lib.zkhip_profile_enable(0)
    return " ".join(f"{k} {st.median(v):.4f}" for k, v in acc.items())
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
for L in logs:
    n = 1 << L
    d = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, d.data_ptr(), stream)); torch.cuda.synchronize()
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    sc = torch.from_numpy(a.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda"); out2 = torch.zeros(12, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    h = C.c_uint64(0); _lib.check(lib.zkhip_prepare_bases_device(d.data_ptr(), n, C.byref(h))); torch.cuda.synchronize()
    t_prep = (time.perf_counter() - t) * 1e3
    hb = C.c_uint64(0); _lib.check(lib.zkhip_prepare_bases_device_c(d.data_ptr(), n, BUCKET_C.get(L, 16), C.byref(hb)))
    run_d = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream))
    run_b = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(hb, 0, sc.data_ptr(), n, out2.data_ptr(), stream))
    md, mind = timed(run_d); mb, minb = timed(run_b)
    same = F.g1_decode_jacobian(out.cpu().numpy().view(np.uint64)) == F.g1_decode_jacobian(out2.cpu().numpy().view(np.uint64))
    print(f"2^{L} device-resident: direct {md:.4f} ms (min {mind:.4f}) | buckets c={BUCKET_C.get(L, 16)} {mb:.4f} ms (min {minb:.4f}) | x{mb / md:.2f} | same point {same} | table build {t_prep:.1f} ms", flush=True)
    print(f"      direct phases: {phases(run_d)}", flush=True)
    print(f"      bucket phases: {phases(run_b)}", flush=True)
    lib.zkhip_release_bases(h); lib.zkhip_release_bases(hb)
    # host buffers, registered (the call an unmodified prover makes)
    hbases = d.cpu().numpy().view(np.uint64).reshape(n, 8).copy()
    ho = np.zeros(12, dtype=np.uint64)
    _lib.check(lib.zkhip_register_bases(hbases.ctypes.data, n))
    run_h = lambda: _lib.check(lib.zkhip_msm_g1(a.ctypes.data, hbases.ctypes.data, n, ho.ctypes.data))
    for _ in range(10): run_h()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); run_h(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"      host buffers, registered: median {st.median(ts):.4f} ms min {min(ts):.4f}", flush=True)
    _lib.check(lib.zkhip_unregister_bases(hbases.ctypes.data))
    del d, sc
