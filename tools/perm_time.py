#!/usr/bin/env python3
"""permutation argument's product columns: zkhip_permutation_products_device (every set in one call) against the composition it replaces
(per set: numerator and denominator row programs, zkhip_fr_grand_product_device, a scaling program for the chaining).
usage: perm_time.py [k] [columns] [chunk_len]"""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
nperm = int(sys.argv[2]) if len(sys.argv) > 2 else 9
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = _lib.load()
dev = torch.device("cuda", 0)
n, u = 1 << k, (1 << k) - 6
rng = random.Random(1)
beta, gamma = rng.randrange(F.R_MOD), rng.randrange(F.R_MOD)


def rand_cols(m):
    a = torch.randint(-(1 << 63), (1 << 63) - 1, (m, n, 4), dtype=torch.int64, device=dev)
    a[:, :, 3] = torch.randint(0, 1 << 61, (m, n), dtype=torch.int64, device=dev)
    return a


V, S = rand_cols(nperm), rand_cols(nperm)
nsets = -(-nperm // chunk)
z = torch.empty((nsets, n, 4), dtype=torch.int64, device=dev)
consts = [F.fr_encode([x])[0] for x in (beta, gamma, E.DELTA, F.omega_for(k))]
vp = (C.c_void_p * nperm)(*[V[i].data_ptr() for i in range(nperm)])
sp = (C.c_void_p * nperm)(*[S[i].data_ptr() for i in range(nperm)])


def one_call():
    _lib.check(lib.zkhip_permutation_products_device(vp, sp, nperm, chunk, k, u, *[c.ctypes.data for c in consts], z.data_ptr(), None))


progs = []
for s in range(nsets):
    lo, hi = s * chunk, min((s + 1) * chunk, nperm)
    progs.append((E.permutation_numerator_program(hi - lo, lo, beta, gamma, k), E.permutation_denominator_program(hi - lo, beta, gamma), lo, hi))
den = torch.empty((n, 4), dtype=torch.int64, device=dev)
z2 = torch.empty((nsets, n, 4), dtype=torch.int64, device=dev)


def per_set():
    last = 1
    for s, (pn, pd, lo, hi) in enumerate(progs):
        pn.run_device([V[c].data_ptr() for c in range(lo, hi)], k, z2[s].data_ptr())
        pd.run_device([V[c].data_ptr() for c in range(lo, hi)] + [S[c].data_ptr() for c in range(lo, hi)], k, den.data_ptr())
        _lib.check(lib.zkhip_fr_grand_product_device(z2[s].data_ptr(), den.data_ptr(), n, z2[s].data_ptr(), None))
        if last != 1:
            sc = E.RowProgram()
            sc.emit(E.OP_MUL, 0, sc.column(0), sc.constant(last))
            sc.run_device([z2[s].data_ptr()], k, z2[s].data_ptr())
        last = F.fr_decode(z2[s, u:u + 1].cpu().numpy().view(np.uint64))[0]            # (already scaled by the earlier sets' last values)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    return best


a, b = timed(one_call), timed(per_set)
same = bool((z[:, :u + 1] == z2[:, :u + 1]).all().item())
print(f"k={k} columns={nperm} chunk={chunk} sets={nsets}: one call {a:.3f} ms, per set {b:.3f} ms, equal={same}")
