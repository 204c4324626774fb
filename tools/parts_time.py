#!/usr/bin/env python3
"""quotient numerator of a wide circuit at few rows: one row program against the sum of `parts` programs (zkhip_fr_eval_rows_sum_device).
usage: parts_time.py [k] [gate_cols] [lookups]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch

from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F

k = int(sys.argv[1]) if len(sys.argv) > 1 else 13
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ek = k + 2
rows = 1 << ek
lib = _lib.load()
dev = torch.device("cuda", 0)
cs = E.halo2_lib_shape(G, NL)
qc = E.quotient_columns(cs)
rng = random.Random(1)
beta, gamma, theta, y = (rng.randrange(1, F.R_MOD) for _ in range(4))
cols = torch.randint(-(1 << 63), (1 << 63) - 1, (qc.total, rows, 4), dtype=torch.int64, device=dev)
cols[:, :, 3] = torch.randint(0, 1 << 61, (qc.total, rows), dtype=torch.int64, device=dev)
ptrs = [cols[i].data_ptr() for i in range(qc.total)]
out = torch.empty((rows, 4), dtype=torch.int64, device=dev)
out2 = torch.empty((rows, 4), dtype=torch.int64, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best


import ctypes as C

cptrs = (C.c_void_p * len(ptrs))(*ptrs)
prog = E.evaluate_h_program(cs, k, ek, beta, gamma, theta, y)
m1 = prog._marshal()                                   # marshalled once: the timed calls are the library's work alone, as from a compiled host
t1 = timed(lambda: _lib.check(lib.zkhip_fr_eval_rows_device(C.byref(m1[0]), cptrs, len(ptrs), ek, 0, C.c_void_p(out.data_ptr()), None)))
print(f"k={k} columns={qc.total} instructions={len(prog.insns)} rows=2^{ek}: one program {t1:.3f} ms")
for parts in (2, 4, 8, 16, 32, 64):
    progs, weights = E.evaluate_h_parts(cs, k, ek, beta, gamma, theta, y, parts)
    ms = [p_._marshal() for p_ in progs]
    arr = (_lib.VmProgram * len(progs))(*[m_[0] for m_ in ms])
    wts = F.fr_encode(list(weights))
    tp = timed(lambda: _lib.check(lib.zkhip_fr_eval_rows_sum_device(arr, wts.ctypes.data, len(progs), cptrs, len(ptrs), ek, C.c_void_p(out2.data_ptr()), None)))
    print(f"  {len(progs):3d} parts: {tp:.3f} ms  equal={bool(torch.equal(out, out2))}  (instructions per part {min(len(p.insns) for p in progs)}..{max(len(p.insns) for p in progs)})")
