#!/usr/bin/env python3
"""Prepared MSM 2^L under scalar distributions that real columns have (bits, bytes, table indices, small negatives, one repeated value):
time and in-library phases.  Looks for cliffs that the uniform bench cannot show."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, fields as F
lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << L
t0m, dm = F.fr_encode([77])[0], F.fr_encode([991])[0]
bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
_lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
h = C.c_uint64(0)
_lib.check(lib.zkhip_prepare_bases_device(bases.data_ptr(), n, C.byref(h)))
out = torch.zeros(12, dtype=torch.int64, device="cuda")

def enc_small(vals):
    """int64 tensor of small non-negative integers -> Montgomery Fr on the device (through a row program: value * R)"""
    raw = torch.zeros((n, 4), dtype=torch.int64, device="cuda"); raw[:, 0] = vals
    from zksnap_circuits_halo2_amd import evaluation as E
    prog = E.RowProgram(); prog.emit(E.OP_MUL, 0, prog.column(0), prog.constant((1 << 256) % F.R_MOD))
    o = torch.empty_like(raw); prog.run_device([raw.data_ptr()], L, o.data_ptr()); torch.cuda.synchronize(); return o

def const(v):
    return torch.from_numpy(F.fr_encode([v % F.R_MOD])[0].view(np.int64)).cuda()

def timed(x, reps=5):
    f = lambda: _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, x.data_ptr(), n, out.data_ptr(), None))
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / reps * 1e3
    lib.zkhip_profile_enable(1); f()
    tm = (C.c_double * 32)(); names = ((C.c_char * 64) * 32)(); k = lib.zkhip_profile_read(tm, names, 32); lib.zkhip_profile_enable(0)
    return "%.3f ms  " % ms + " ".join(f"{names[i].value.decode()}={tm[i]:.3f}" for i in range(k))

g = torch.Generator(device="cuda"); g.manual_seed(5)
uni = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g); uni[:, 3] &= (1 << 61) - 1
rows = torch.arange(n, device="cuda")
cases = {
    "uniform": uni,
    "bits (random 0/1)": enc_small(torch.randint(0, 2, (n,), device="cuda", generator=g)),
    "bytes (random 0..255)": enc_small(torch.randint(0, 256, (n,), device="cuda", generator=g)),
    "table column (row index)": enc_small(rows),
    "all equal to -1": const(-1).repeat(n, 1).contiguous(),
}
x = uni.clone(); x[::20] = const(-1); cases["uniform, 5% equal to -1"] = x
x = uni.clone(); x[::4] = const(-1); x[1::4] = const(1); x[2::4] = const(2); cases["uniform 25%, -1 / 1 / 2 25% each"] = x
x = enc_small(torch.randint(0, 1 << 16, (n,), device="cuda", generator=g)); x[rows % 3 == 0] = 0; cases["16-bit values, a third zero"] = x
for name, x in cases.items():
    print(f"2^{L} {name:36s} {timed(x.contiguous())}", flush=True)
