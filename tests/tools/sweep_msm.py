#!/usr/bin/env python3
"""MSM size sweep (prepared and general paths), uniform and witness-like scalars; checks every result against the
structured identity MSM(a, (t0+i d)G) = [sum a_i (t0+i d)] G using the C oracle for the scalar sum."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from oracle import cpu_ref as Cr
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream
T0, D = 0x5A4B534E41500002, 0x9E3779B97F4A7C15F39CC0605CEDC835
t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
logs = [int(x) for x in sys.argv[1:]] or [10, 13, 15, 16, 18, 20, 22, 24]
print("log_n kind   prepared_ms Mpts/s   general_ms Mpts/s  check")
for L in logs:
    n = 1 << L
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), stream))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
    for kind in (0, 1):
        sc_h = Cr.gen_scalars(1000 + L, n, kind)
        sc = torch.from_numpy(sc_h.view(np.int64)).cuda()
        out = torch.zeros(12, dtype=torch.int64, device="cuda"); out2 = torch.zeros(12, dtype=torch.int64, device="cuda")
        reps = 10 if L <= 20 else 3
        tp = timed(lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, sc.data_ptr(), n, out.data_ptr(), stream)), reps)
        tg = timed(lambda: _lib.check(lib.zkhip_msm_g1_device(sc.data_ptr(), bases.data_ptr(), n, out2.data_ptr(), stream)), reps)
        exp = Cr.jac_to_affine(Cr.scalar_mul(Cr.expected_scalar(sc_h, T0, D), Cr.generator()))
        ok = all(np.array_equal(Cr.jac_to_affine(np.ascontiguousarray(o.cpu().numpy().view(np.uint64))), exp) for o in (out, out2))
        print(f"{L:5d} {'uniform' if kind == 0 else 'witness':7s} {tp:10.3f} {n/tp/1e3:8.1f} {tg:10.3f} {n/tg/1e3:8.1f}  {'OK' if ok else 'MISMATCH'}", flush=True)
    lib.zkhip_release_bases(h)
    del bases
