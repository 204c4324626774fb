#!/usr/bin/env python3
"""First-light GPU checks against the Python big-int oracle (later superseded by tests/)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from oracle import bn254 as O
import zksnap_circuits_halo2_amd as Z
from zksnap_circuits_halo2_amd import _lib, fields as F

lib = _lib.load()
buf = (b" " * 256)
import ctypes
nm = ctypes.create_string_buffer(256); _lib.check(lib.zkhip_device_name(nm, 256)); print("device:", nm.value.decode())
rng = O.SplitMix64(1234)
fails = 0
def check(name, ok):
    global fails
    print(("PASS " if ok else "FAIL ") + name, flush=True)
    if not ok: fails += 1

# ---- field ops
for field, p in ((0, O.Q_MOD), (1, O.R_MOD)):
    n = 4096
    a = [rng.fr() % p for _ in range(n)]; b = [rng.fr() % p for _ in range(n)]
    edge = [0, 1, p - 1, p - 2, 2, (p - 1) // 2]
    for i, e in enumerate(edge): a[i] = e; b[i] = edge[(i * 5 + 1) % len(edge)]
    enc = lambda v: np.array([O.limbs4(O.to_mont(x, p)) for x in v], dtype=np.uint64)
    A, B = enc(a), enc(b)
    for op, f in ((0, lambda x, y: x * y % p), (1, lambda x, y: (x + y) % p), (2, lambda x, y: (x - y) % p), (3, lambda x, y: x * x % p)):
        out = np.zeros_like(A)
        _lib.check(lib.zkhip_test_field_op(field, op, A.ctypes.data, B.ctypes.data, out.ctypes.data, n))
        exp = enc([f(x, y) for x, y in zip(a, b)])
        check(f"field {field} op {op}", np.array_equal(out, exp))

# ---- curve ops
n = 512
ks = [rng.fr() for _ in range(n)]
G = O.G1_GEN
pts = [O.to_jac(O.scalar_mul(k, G)) for k in ks[:64]]
# extend cheaply by additions
while len(pts) < n: pts.append(O.jac_add(pts[-1], pts[len(pts) - 64]))
aff = O.batch_to_affine(pts)
aff2 = aff[1:] + aff[:1]
aff[5] = None; aff2[9] = None; aff[11] = None; aff2[11] = None
aff2[20] = aff[20]            # doubling through madd
aff2[21] = O.neg(aff[21])     # P + (-P)
A, B = F.g1_encode(aff), F.g1_encode(aff2)
for op, f in ((0, lambda P, Q: O.add(P, Q)), (1, lambda P, Q: O.add(P, P)), (2, lambda P, Q: O.add(P, O.neg(Q)))):
    out = np.zeros((n, 12), dtype=np.uint64)
    _lib.check(lib.zkhip_test_g1_op(op, A.ctypes.data, B.ctypes.data, out.ctypes.data, n))
    got = [F.g1_decode_jacobian(out[i]) for i in range(n)]
    exp = [f(P, Q) for P, Q in zip(aff, aff2)]
    bad = [i for i in range(n) if got[i] != exp[i]]
    check(f"g1 op {op}" + (f" first bad {bad[:5]}" if bad else ""), not bad)

# ---- MSM
def make_bases(n, seed):
    g = O.SplitMix64(seed)
    t0, d = g.fr(), g.fr()
    P, D = O.to_jac(O.scalar_mul(t0, G)), O.to_jac(O.scalar_mul(d, G))
    js = []
    for _ in range(n):
        js.append(P); P = O.jac_add(P, D)
    return O.batch_to_affine(js), [(t0 + i * d) % O.R_MOD for i in range(n)]

for n in (0, 1, 2, 3, 5, 31, 32, 100, 1000, 4096, 1 << 14):
    t = time.time()
    bases, ts = make_bases(n, 77 + n)
    sc = [rng.fr() for _ in range(n)]
    if n >= 100:
        sc[3] = 0; sc[4] = 1; sc[5] = O.R_MOD - 1; sc[6] = (1 << 253); sc[7] = sc[8]
        bases[9] = None
        ts[9] = 0
    exp = O.scalar_mul(sum(s * t_ for s, t_ in zip(sc, ts)) % O.R_MOD, G) if n else None
    got = F.g1_decode_jacobian(Z.best_multiexp(F.fr_encode(sc), F.g1_encode(bases)))
    check(f"msm n={n} ({time.time()-t:.1f}s)", got == exp)
    if n in (31, 100):
        check(f"msm n={n} vs naive", got == O.msm_naive(sc, bases))

# witness-like + repeated bases (heavy buckets)
n = 1 << 13
bases, ts = make_bases(64, 5)
bases = (bases * (n // 64)); ts = ts * (n // 64)
sc = O.witness_like_scalars(99, n)
for i in range(0, n, 3): sc[i] = 1
exp = O.scalar_mul(sum(s * t_ for s, t_ in zip(sc, ts)) % O.R_MOD, G)
got = F.g1_decode_jacobian(Z.best_multiexp(F.fr_encode(sc), F.g1_encode(bases)))
check("msm witness-like, repeated bases", got == exp)

# ---- NTT
for L in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 13):
    N = 1 << L
    a = [rng.fr() for _ in range(N)]
    w = O.omega_for(L)
    exp = O.best_fft(a, w, L)
    if L <= 8: assert exp == O.dft_naive(a, w)
    arr = F.fr_encode(a)
    Z.best_fft(arr, F.fr_encode([w])[0], L)
    check(f"ntt L={L}", F.fr_decode(arr) == exp)

dom = Z.EvaluationDomain(4, 6)
od = O.EvaluationDomain(4, 6)
a = [rng.fr() for _ in range(dom.n)]
A = F.fr_encode(a)
check("lagrange_to_coeff", F.fr_decode(dom.lagrange_to_coeff(A)) == od.lagrange_to_coeff(a))
ext = dom.coeff_to_extended(A)
check("coeff_to_extended", F.fr_decode(ext) == od.coeff_to_extended(a))
check("divide_by_vanishing", F.fr_decode(dom.divide_by_vanishing_poly(ext)) == od.divide_by_vanishing_poly(F.fr_decode(ext)))
check("extended_to_coeff", F.fr_decode(dom.extended_to_coeff(ext)) == od.extended_to_coeff(F.fr_decode(ext)))
print("FAILS:", fails)
sys.exit(1 if fails else 0)
