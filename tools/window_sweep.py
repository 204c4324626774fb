#!/usr/bin/env python3
"""Prepared MSM under every wide window size, same box, alternating (round 4, review item 3).
  window_sweep.py [logs=20,21,22,23] [cs=16,17,18,19,20] [reps=40] [rounds=3]
For each size: one set of bases, one prepared table per window size (zkhip_prepare_bases_device_c), uniform and witness-like scalars
(60 % zero, 30 % below 2^88, 10 % uniform); the variants are timed in alternating rounds of `reps` calls, the results of all window sizes must be
the same group element (the automatic choice among them is what the parity tests pin), and the median per-phase times (in-library HIP events) are printed next to the
end-to-end median.  Output lines are meant to be committed as profiles/r04_window_sweep.txt."""
import os, sys, ctypes as C, statistics as st
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,21,22,23").split(",")]
cs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,17,18,19,20").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
stream = torch.cuda.current_stream().cuda_stream
T0, DD = 77, 991


def scalars(n, kind, rng):
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] = rng.integers(0, 0x30644E72E131A029, size=n, dtype=np.uint64)
    if kind == "witness":
        sel = rng.integers(0, 10, size=n)
        a[sel < 6] = 0
        # 30 %: integers below 2^88, built as plain limbs (lo, mid & 2^24 - 1, 0, 0) and brought into the Montgomery memory format by the
        # library's field-multiply hook: mul(x, R^2) = x R
        small = np.flatnonzero((sel >= 6) & (sel < 9))
        a[small, 2:] = 0
        a[small, 1] &= np.uint64((1 << 24) - 1)
        r2 = F.fr_encode([pow(2, 256, R)])[0]            # Montgomery form of R: mul(x_plain_words, mont(R)) = x * R * R / R = x R = Montgomery(x)
        xs = np.ascontiguousarray(a[small])
        b = np.ascontiguousarray(np.broadcast_to(r2, xs.shape))
        out = np.empty_like(xs)
        _lib.check(lib.zkhip_test_field_op(1, 0, xs.ctypes.data, b.ctypes.data, out.ctypes.data, len(xs)))
        a[small] = out
    return a


print(f"# window sweep: reps {reps} x rounds {rounds}, alternating per round; device {torch.cuda.get_device_name(0)}")
for L in logs:
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    t0, dd = F.fr_encode([T0])[0], F.fr_encode([DD])[0]
    _lib.check(lib.zkhip_g1_gen_walk_device(t0.ctypes.data, dd.ctypes.data, n, bases.data_ptr(), stream))
    handles = {}
    for c in cs:
        h = C.c_uint64(0)
        _lib.check(lib.zkhip_prepare_bases_device_c(bases.data_ptr(), n, c, C.byref(h)))
        handles[c] = h
    ha = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(ha)))
    auto = lib.zkhip_prepared_window_bits(ha)
    lib.zkhip_release_bases(ha)
    for kind in ("uniform", "witness"):
        rng = np.random.default_rng(1000 + L)
        a = scalars(n, kind, rng)
        sc = torch.from_numpy(a.view(np.int64)).cuda()
        outs = {c: torch.zeros(12, dtype=torch.int64, device="cuda") for c in cs}
        run = lambda c: _lib.check(lib.zkhip_msm_g1_prepared_device(handles[c], 0, sc.data_ptr(), n, outs[c].data_ptr(), stream))
        for c in cs:
            for _ in range(5): run(c)
        torch.cuda.synchronize()
        pts = {c: F.g1_decode_jacobian(outs[c].cpu().numpy().view(np.uint64)) for c in cs}
        same = len(set(pts.values())) == 1
        ts = {c: [] for c in cs}
        for _ in range(rounds):
            for c in cs:
                for _ in range(10): run(c)        # keep the clocks up across the switch
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(c); e1.record(); e1.synchronize(); ts[c].append(e0.elapsed_time(e1))
        ms = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)()
        lib.zkhip_profile_enable(1)
        ph = {c: {} for c in cs}
        for _ in range(12):
            for c in cs:
                run(c); torch.cuda.synchronize()
                k = lib.zkhip_profile_read(ms, names, 32)
                per = {}
                for i in range(k): per[names[i].value.decode()] = per.get(names[i].value.decode(), 0.0) + ms[i]
                for nm, v in per.items(): ph[c].setdefault(nm, []).append(v)
        lib.zkhip_profile_enable(0)
        best = min(cs, key=lambda c: st.median(ts[c]))
        for c in cs:
            med = st.median(ts[c])
            print(f"2^{L} {kind:8s} c={c:2d} median {med:.4f} ms min {min(ts[c]):.4f} ({n / med / 1e3:7.1f} Mpoints/s){' *best*' if c == best else ''}"
                  f"{' [auto]' if c == auto else ''} results_equal={same} | " + " ".join(f"{k} {st.median(v):.4f}" for k, v in ph[c].items()), flush=True)
        del sc
    for c in cs: lib.zkhip_release_bases(handles[c])
    del bases
    torch.cuda.empty_cache()
