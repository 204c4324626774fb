#!/usr/bin/env python3
"""Time the fused quotient-numerator pass (zkhip_fr_eval_rows) on the wrapper-circuit shape: halo2-lib BaseConfig with
k = 22, advice [4], lookup [1,0,0], fixed 1 (/root/reference/aggregator/benches/wrapper_circuit.rs:61-68), degree 4 ->
extended_k = k + 2, 7 permutation columns in 4 sets.  `quotient_time.py [extended_k ...]` (default 20 22 24)."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F
lib = _lib.load()
stream = torch.cuda.current_stream().cuda_stream

def wrapper_cs():
    gates = [[E.Fixed(i) * (E.Advice(i, 0) + E.Advice(i, 1) * E.Advice(i, 2) - E.Advice(i, 3))] for i in range(4)]
    lookups = [E.Lookup([E.Advice(4)], [E.Fixed(5)])]
    perm = [("advice", i) for i in range(5)] + [("fixed", 4), ("instance", 0)]
    return E.ConstraintSystem(num_fixed=6, num_advice=5, num_instance=1, gates=gates, lookups=lookups, permutation_columns=perm,
                              blinding_factors=5, degree=4)

cs = wrapper_cs()
qc = E.quotient_columns(cs)
rng = random.Random(1)
beta, gamma, theta, y = (rng.randrange(F.R_MOD) for _ in range(4))
for ek in [int(x) for x in sys.argv[1:]] or [20, 22, 24]:
    k = ek - 2
    rows = 1 << ek
    prog = E.evaluate_h_program(cs, k, ek, beta, gamma, theta, y)
    reads = {(o[1], o[2]) for ins in prog.insns for o in ins[2:5] if o[0] == E.SRC_COLUMN}
    n_mul = sum(1 for i in prog.insns if i[0] in (E.OP_MUL, E.OP_SQR, E.OP_MAD))
    cols = []
    for _ in range(qc.total):
        t = torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device="cuda")
        t[:, 3] &= (1 << 61) - 1            # canonical: below the modulus' top limb
        cols.append(t)
    out = torch.zeros(rows * 4, dtype=torch.int64, device="cuda")
    ptrs = [t.data_ptr() for t in cols]
    run = lambda: prog.run_device(ptrs, ek, out.data_ptr(), stream=stream)
    run(); torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    byts = (len(reads) + 1) * 32 * rows
    print(f"extended_k={ek} rows=2^{ek} columns={qc.total} insns={len(prog.insns)} (mul-type {n_mul}) column reads/row={len(reads)}  "
          f"{ms:8.3f} ms  {rows/ms/1e3:8.1f} Mrows/s  algorithmic {byts/1e9:.2f} GB -> {byts/ms/1e6:7.1f} GB/s ({byts/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
    del cols, out
