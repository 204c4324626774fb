#!/usr/bin/env python3
"""Randomised differential run of the C ABI against the oracles (test infrastructure: imports oracle/).  `fuzz_parity.py SECONDS [SEED] [large]`:
random sizes, window overrides, scalar distributions and programs until the time is up; stops at the first mismatch with a repro line."""
import ctypes as C, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from oracle import bn254 as O, cpu_ref as Cr
from zksnap_circuits_halo2_amd import _lib, arithmetic as A, evaluation as E, fields as F

Cr.load()
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"      # MSM cases of 2^20 .. 5 * 2^20 points only
rng = random.Random(seed0)
R = O.R_MOD
T0, D = 0x5A4B534E41500999, 0x9E3779B97F4A7C15F39CC0605CEDC835
t0m, dm = F.fr_encode([T0])[0], F.fr_encode([D])[0]
counts = {}

def aff(x): return Cr.jac_to_affine(np.ascontiguousarray(x))

def scalars(n, seed):
    kind = rng.randrange(6)
    if kind == 5:                                      # bytes on the edges of the 8-bit signed recoding of the direct tables (carry ripples)
        pick = np.random.default_rng(seed)
        raw = pick.choice(np.array([0x00, 0x7f, 0x80, 0x81, 0xff, 0x01], dtype=np.uint8), size=(n, 32), p=[0.15, 0.25, 0.25, 0.1, 0.2, 0.05])
        raw[:, 31] &= 0x1f                             # below 2^253 < r: canonical integers
        vals = [int.from_bytes(bytes(r_), "little") for r_ in raw[: min(n, 4096)]]
        enc = F.fr_encode(vals)
        return np.ascontiguousarray(enc[np.arange(n) % enc.shape[0]])
    if kind < 2: return Cr.gen_scalars(seed, n, kind)
    if kind == 4:                                      # what real columns look like: uniform values with clusters of 1, -1, small constants
        s = Cr.gen_scalars(seed, n, 0)
        consts = F.fr_encode([1, R - 1, 2, rng.randrange(1 << 16), R - rng.randrange(1, 1 << 16)])
        pick = np.random.default_rng(seed + 1)
        mask = pick.random(n) < rng.choice([0.05, 0.3, 0.9])
        s[mask] = consts[pick.integers(0, consts.shape[0], int(mask.sum()))]
        return np.ascontiguousarray(s)
    if kind == 2:                                      # few distinct values: heavy buckets
        vals = Cr.gen_scalars(seed, rng.randint(1, 4), 0)
        return np.ascontiguousarray(vals[np.random.default_rng(seed).integers(0, vals.shape[0], n)])
    s = Cr.gen_scalars(seed, n, 0); s[rng.randrange(n):] = 0; return s      # zero tail

def fuzz_msm(seed):
    n = rng.choice([rng.randint(1, 300), rng.randint(300, 20000), rng.randint(20000, 300000)])
    if LARGE: n = rng.randint(1 << 20, 5 << 20)      # the wide-window path with ragged sizes
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    sc = scalars(n, seed)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    exp = aff(Cr.scalar_mul(Cr.expected_scalar(sc, T0, D), Cr.generator()))
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    c = rng.choice([0, 0, rng.randint(2, 16)])
    _lib.check(lib.zkhip_msm_g1_device_c(dsc.data_ptr(), bases.data_ptr(), n, out.data_ptr(), c, None))
    torch.cuda.synchronize()
    assert np.array_equal(aff(out.cpu().numpy().view(np.uint64)), exp), f"general msm n={n} c={c}"
    cp = rng.choice([0, 0, rng.randint(2, 20)])
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device_c(bases.data_ptr(), n, cp, C.byref(h)))
    try:
        off = rng.randrange(0, max(1, n // 3)); m = rng.randint(1, n - off)
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, 0, dsc.data_ptr(), n, out.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(aff(out.cpu().numpy().view(np.uint64)), exp), f"prepared msm n={n} c={cp}"
        _lib.check(lib.zkhip_msm_g1_prepared_device(h, off, dsc.data_ptr() + off * 32, m, out.data_ptr(), None))
        torch.cuda.synchronize()
        sub = np.ascontiguousarray(sc[off:off + m])
        e2 = aff(Cr.scalar_mul(Cr.expected_scalar(sub, (T0 + off * D) % R, D), Cr.generator()))
        assert np.array_equal(aff(out.cpu().numpy().view(np.uint64)), e2), f"prepared sub-range n={n} c={cp} off={off} m={m}"
    finally:
        lib.zkhip_release_bases(h)

def fuzz_ntt(seed):
    L = rng.choice([rng.randint(0, 17), rng.randint(9, 21)])      # (one / two / three passes; tiles of 1024 elements from L = 10)
    a = Cr.gen_scalars(seed, 1 << L, rng.randrange(2))
    om = F.fr_encode([F.omega_for(L)])[0]
    ref = a.copy(); Cr.best_fft(ref, om, L, 4)
    got = a.copy(); A.best_fft(got, om, L)
    assert np.array_equal(got, ref), f"ntt L={L}"

def fuzz_poly(seed):
    n = rng.choice([rng.randint(1, 70), rng.randint(70, 5000), rng.randint(5000, 200000)])
    a = Cr.gen_scalars(seed, n, rng.randrange(2)); x = Cr.gen_scalars(seed + 1, 1, 0)[0]
    assert np.array_equal(A.eval_polynomial(a, x), Cr.eval_polynomial(a, x)), f"eval n={n}"
    assert np.array_equal(A.kate_division(a, x), Cr.kate_division(a, x)), f"kate n={n}"
    assert np.array_equal(A.prefix_product(a), Cr.prefix_product(a)), f"prefix n={n}"
    b, r = a.copy(), a.copy(); A.batch_invert(b); Cr.batch_invert(r)
    assert np.array_equal(b, r), f"invert n={n}"

def fuzz_rows(seed):
    log_rows = rng.randint(0, 8); rows = 1 << log_rows; n_cols = rng.randint(1, 4)
    cols = [[rng.randrange(R) if rng.random() < 0.9 else rng.choice([0, 1, R - 1]) for _ in range(rows)] for _ in range(n_cols)]
    omega = O.omega_for(log_rows) if rng.random() < 0.5 else None
    prev = [rng.randrange(R) for _ in range(rows)] if rng.random() < 0.5 else None
    p = E.RowProgram(rot_scale=rng.choice([1, 2, 4]), omega=omega)
    written = []
    def operand():
        kinds = ["const", "col"] + (["reg"] if written else []) + (["prev"] if prev is not None else []) + (["rowpow"] if omega else [])
        k = rng.choice(kinds)
        if k == "const": return p.constant(rng.choice([0, 1, R - 1, rng.randrange(R)]))
        if k == "col": return p.column(rng.randrange(n_cols), rng.choice([0, 1, -1, 2, -5]))
        if k == "reg": return E.RowProgram.reg(rng.choice(written))
        return E.RowProgram.PREV if k == "prev" else E.RowProgram.ROWPOW
    n_regs = rng.choice([6, 8, 12, 16])
    for _ in range(rng.randint(1, 120)):
        dst = rng.randrange(n_regs); p.emit(rng.randrange(8), dst, operand(), operand(), operand())
        if dst not in written: written.append(dst)
    p.result_reg = rng.choice(written)
    out = F.fr_encode(prev) if prev is not None else None
    got = F.fr_decode(p.run([F.fr_encode(c) for c in cols] + [F.fr_encode(cols[0])] * (p.n_columns - n_cols), log_rows, out=out, accumulate=prev is not None))
    exp = O.row_program_run(p.insns, p.constants, p.rotations, p.rot_scale, p.result_reg, cols + [cols[0]] * 8, log_rows, omega=omega, prev=prev)
    assert got == exp, f"row program rows={rows}"

def fuzz_lookup(seed):
    n = rng.randint(1, 3000); bits = rng.choice([1, 3, 8, 250])
    table = [rng.randrange(1 << bits) % R for _ in range(n)]
    usable = rng.randint(1, n)
    inputs = [rng.choice(table[:usable]) for _ in range(n)]
    gi, gt = E.permute_expression_pair(F.fr_encode(inputs), F.fr_encode(table), usable)
    ei, et = O.permute_expression_pair(inputs, table, usable)
    assert F.fr_decode(gi) == ei and F.fr_decode(gt) == et, f"lookup n={n} usable={usable} bits={bits}"

def fuzz_batched(seed):
    """the batched / strided device entry points against their single-call forms"""
    import zksnap_circuits_halo2_amd as Z
    k = rng.randint(1, 12); n = 1 << k; B = rng.randint(1, 5); pad = rng.choice([0, 8, 40])
    dom = Z.EvaluationDomain(4, k)
    polys = [Cr.gen_scalars(seed + b, n, rng.randrange(2)) for b in range(B)]
    stride = n + pad
    x = torch.zeros((B, stride, 4), dtype=torch.int64, device="cuda")
    for b in range(B): x[b, :n] = torch.from_numpy(polys[b].view(np.int64)).cuda()
    om = F.fr_encode([F.omega_for(k)])[0]
    _lib.check(lib.zkhip_ntt_fr_batch_device(x.data_ptr(), om.ctypes.data, k, B, stride, None))
    torch.cuda.synchronize()
    for b in range(B):
        ref = polys[b].copy(); Cr.best_fft(ref, om, k, 4)
        assert np.array_equal(x[b, :n].cpu().numpy().view(np.uint64), ref), f"ntt batch k={k} B={B} pad={pad}"
    assert not x[:, n:].any().item(), "ntt batch wrote into the padding"
    # coset round trip, device, strided
    en, ek = dom.extended_len(), dom.extended_k
    c = torch.zeros((B, stride, 4), dtype=torch.int64, device="cuda")
    for b in range(B): c[b, :n] = torch.from_numpy(polys[b].view(np.int64)).cuda()
    e = torch.zeros((B, en + pad, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_coeff_to_extended_device(c.data_ptr(), stride, k, e.data_ptr(), en + pad, ek, B, dom.extended_omega.ctypes.data, dom.g_coset.ctypes.data, None))
    torch.cuda.synchronize()
    assert np.array_equal(e[B - 1, :en].cpu().numpy().view(np.uint64), dom.coeff_to_extended(polys[B - 1])), f"coeff_to_extended batch k={k}"
    o = torch.zeros((B, 3 * n + pad, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_extended_to_coeff_device(e.data_ptr(), en + pad, ek, dom.extended_omega_inv.ctypes.data, dom.extended_ifft_divisor.ctypes.data,
                                                  dom.g_coset.ctypes.data, o.data_ptr(), 3 * n + pad, 3 * n, B, None))
    torch.cuda.synchronize()
    for b in range(B):
        back = o[b].cpu().numpy().view(np.uint64)
        assert np.array_equal(back[:n], polys[b]) and not back[n:].any(), f"extended_to_coeff batch k={k} b={b}"
    # batched evaluation and batched prepared MSM
    pt = Cr.gen_scalars(seed + 99, 1, 0)[0]
    ev = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    ptrs = (C.c_void_p * B)(*[c[b].data_ptr() for b in range(B)])
    _lib.check(lib.zkhip_fr_eval_polynomial_batch_device(ptrs, B, n, pt.ctypes.data, ev.data_ptr(), None))
    torch.cuda.synchronize()
    for b in range(B):
        assert np.array_equal(ev[b].cpu().numpy().view(np.uint64), Cr.eval_polynomial(polys[b], pt)), f"eval batch k={k} b={b}"
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    h = C.c_uint64(0)
    _lib.check(lib.zkhip_prepare_bases_device_c(bases.data_ptr(), n, rng.choice([0, rng.randint(2, 16)]), C.byref(h)))
    try:
        outs = torch.zeros((B, 12), dtype=torch.int64, device="cuda")
        _lib.check(lib.zkhip_msm_g1_prepared_batch_device(h, 0, c.data_ptr(), n, B, stride, outs.data_ptr(), None))
        torch.cuda.synchronize()
        for b in range(B):
            exp = aff(Cr.scalar_mul(Cr.expected_scalar(polys[b], T0, D), Cr.generator()))
            assert np.array_equal(aff(outs[b].cpu().numpy().view(np.uint64)), exp), f"msm batch k={k} b={b}"
    finally:
        lib.zkhip_release_bases(h)

def fuzz_sharded(seed):
    """the host-buffer MSM over registered bases cut into a random number of shards (the multi-GPU path on one card), random sub-ranges"""
    n = rng.choice([rng.randint(1, 500), rng.randint(500, 60000)])
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    torch.cuda.synchronize()
    hb = np.ascontiguousarray(bases.cpu().numpy().view(np.uint64).reshape(n, 8))
    sc = scalars(n, seed)
    S = rng.choice([1, 2, 3, 5, 8, 13])
    _lib.check(lib.zkhip_set_msm_shards(S))
    _lib.check(lib.zkhip_register_bases(hb.ctypes.data, n))
    try:
        for _ in range(3):
            off = rng.randrange(0, n); m = rng.randint(1, n - off)
            sub = np.ascontiguousarray(sc[off:off + m])
            out = np.zeros(12, dtype=np.uint64)
            _lib.check(lib.zkhip_msm_g1(sub.ctypes.data, hb[off:].ctypes.data, m, out.ctypes.data))
            exp = aff(Cr.scalar_mul(Cr.expected_scalar(sub, (T0 + off * D) % R, D), Cr.generator()))
            assert np.array_equal(aff(out), exp), f"sharded msm n={n} S={S} off={off} m={m}"
            # the same range with the scalars resident in HBM (round 3: fans out over the shards), alone and as a batch with a padded stride
            dsub = torch.from_numpy(sub.view(np.int64)).cuda()
            dout = torch.zeros(12, dtype=torch.int64, device="cuda")
            _lib.check(lib.zkhip_msm_g1_registered_device(hb[off:].ctypes.data, dsub.data_ptr(), m, dout.data_ptr(), None))
            torch.cuda.synchronize()
            assert np.array_equal(aff(dout.cpu().numpy().view(np.uint64)), exp), f"sharded device-resident msm n={n} S={S} off={off} m={m}"
            Bt, pad = rng.randint(1, 4), rng.choice([0, 3, 16])
            vecs = [np.ascontiguousarray(scalars(n, seed + 7 * b + 1)[off:off + m]) for b in range(Bt)]
            dv = torch.zeros((Bt, m + pad, 4), dtype=torch.int64, device="cuda")
            for b in range(Bt): dv[b, :m] = torch.from_numpy(vecs[b].view(np.int64)).cuda()
            douts = torch.zeros((Bt, 12), dtype=torch.int64, device="cuda")
            _lib.check(lib.zkhip_msm_g1_registered_batch_device(hb[off:].ctypes.data, dv.data_ptr(), m, Bt, m + pad, douts.data_ptr(), None))
            torch.cuda.synchronize()
            for b in range(Bt):
                eb = aff(Cr.scalar_mul(Cr.expected_scalar(vecs[b], (T0 + off * D) % R, D), Cr.generator()))
                assert np.array_equal(aff(douts[b].cpu().numpy().view(np.uint64)), eb), f"sharded device-resident batch n={n} S={S} off={off} m={m} b={b}/{Bt}"
    finally:
        lib.zkhip_unregister_bases(hb.ctypes.data)
        lib.zkhip_set_msm_shards(0)

def fuzz_host_chunked(seed):
    """large only: the host-buffer MSM over a registered array of 2^21 .. 5 * 2^20 points -- the chunked upload (pieces accumulated into one bucket
    set) -- on a random sub-range, with 1 .. 3 shards"""
    n = rng.randint(1 << 21, 5 << 20)
    bases = torch.empty(n * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_g1_gen_walk_device(t0m.ctypes.data, dm.ctypes.data, n, bases.data_ptr(), None))
    torch.cuda.synchronize()
    hb = np.ascontiguousarray(bases.cpu().numpy().view(np.uint64).reshape(n, 8))
    del bases
    sc = scalars(n, seed)
    S = rng.choice([1, 1, 2, 3])
    _lib.check(lib.zkhip_set_msm_shards(S))
    _lib.check(lib.zkhip_register_bases(hb.ctypes.data, n))
    try:
        for off, m in ((0, n), (rng.randrange(0, n // 4), rng.randint(n // 2, n - n // 4))):
            sub = np.ascontiguousarray(sc[off:off + m])
            out = np.zeros(12, dtype=np.uint64)
            _lib.check(lib.zkhip_msm_g1(sub.ctypes.data, hb[off:].ctypes.data, m, out.ctypes.data))
            exp = aff(Cr.scalar_mul(Cr.expected_scalar(sub, (T0 + off * D) % R, D), Cr.generator()))
            assert np.array_equal(aff(out), exp), f"chunked host msm n={n} S={S} off={off} m={m}"
    finally:
        lib.zkhip_unregister_bases(hb.ctypes.data)
        lib.zkhip_set_msm_shards(0)

_g2_walk = {}
def fuzz_g2(seed):
    """G2 MSM on a prefix of a fixed walk of G2 points (the oracle builds the walk once: big-integer Fq2 arithmetic)"""
    if not _g2_walk:
        P, Dp = O.g2_to_jac(O.g2_scalar_mul(T0, O.G2_GEN)), O.g2_to_jac(O.g2_scalar_mul(D, O.G2_GEN))
        pts = []
        for _ in range(1500):
            pts.append(P); P = O.g2_jac_add(P, Dp)
        _g2_walk["enc"] = np.array([O.g2_affine_to_limbs(O.g2_to_affine(J)) for J in pts], dtype=np.uint64).reshape(-1, 16)
    n = rng.randint(1, 1500)
    sc = scalars(n, seed)
    got = O.g2_jac_from_limbs([int(x) for x in A.best_multiexp_g2(sc, _g2_walk["enc"][:n])])
    assert got == O.g2_scalar_mul(Cr.expected_scalar(sc, T0, D), O.G2_GEN), f"g2 msm n={n}"

def fuzz_perm(seed):
    """zkhip_permutation_products (every set of the permutation argument in one call) against the products written out with big integers"""
    k = rng.randint(0, 9); n = 1 << k; nperm = rng.randint(1, 9); chunk = rng.randint(1, 4); usable = rng.randint(0, n)
    beta, gamma = rng.randrange(1, R), rng.randrange(R)
    vals = [[rng.randrange(R) if rng.random() < 0.9 else rng.choice([0, 1, R - 1]) for _ in range(n)] for _ in range(nperm)]
    sig = [[rng.randrange(R) for _ in range(n)] for _ in range(nperm)]
    if rng.random() < 0.2 and usable:                  # a zero denominator: every later value of every later set is zero (BatchInvert's convention)
        c, i = rng.randrange(nperm), rng.randrange(usable)
        sig[c][i] = (-(vals[c][i] + gamma)) * pow(beta, -1, R) % R
    omega = O.omega_for(k)
    exp, last = [], 1
    for lo in range(0, nperm, chunk):
        z = [last]
        for i in range(n - 1):
            if i < usable:
                a = b = 1
                for c in range(lo, min(lo + chunk, nperm)):
                    a = a * (vals[c][i] + pow(O.FR_DELTA, c, R) * beta % R * pow(omega, i, R) + gamma) % R
                    b = b * (vals[c][i] + beta * sig[c][i] + gamma) % R
                z.append(z[-1] * a % R * pow(b, -1, R) % R if b else 0)
            else:
                z.append(z[-1])
        if usable < n:
            last = z[usable]
        else:                                            # every row is usable: the next set starts at the product over all n rows
            a = b = 1
            for c in range(lo, min(lo + chunk, nperm)):
                a = a * (vals[c][n - 1] + pow(O.FR_DELTA, c, R) * beta % R * pow(omega, n - 1, R) + gamma) % R
                b = b * (vals[c][n - 1] + beta * sig[c][n - 1] + gamma) % R
            last = z[-1] * a % R * pow(b, -1, R) % R if b else 0
        exp += z
    V, S = [F.fr_encode(c) for c in vals], [F.fr_encode(c) for c in sig]
    consts = [F.fr_encode([x])[0] for x in (beta, gamma, O.FR_DELTA, omega)]
    nsets = -(-nperm // chunk)
    z = np.zeros((nsets * n, 4), dtype=np.uint64)
    vp, sp = (C.c_void_p * nperm)(*[a.ctypes.data for a in V]), (C.c_void_p * nperm)(*[a.ctypes.data for a in S])
    _lib.check(lib.zkhip_permutation_products(vp, sp, nperm, chunk, k, usable, *[c.ctypes.data for c in consts], z.ctypes.data))
    assert F.fr_decode(z) == exp, f"permutation products k={k} columns={nperm} chunk={chunk} usable={usable}"

def fuzz_lincomb(seed):
    """zkhip_fr_linear_combination_device (one group, several groups, more than 64 groups) against big integers; the output may be a column"""
    log_n = rng.randint(0, 9); n = 1 << log_n
    count = rng.choice([rng.randint(0, 3), rng.randint(4, 70), rng.randint(70, 2200)])
    base = [[rng.randrange(R) if rng.random() < 0.9 else rng.choice([0, 1, R - 1]) for _ in range(n)] for _ in range(min(max(count, 1), 12))]
    pick = [rng.randrange(len(base)) for _ in range(count)]
    coeffs = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(count)]
    d_base = [torch.from_numpy(F.fr_encode(c).view(np.int64)).cuda() for c in base]
    alias = count > 0 and rng.random() < 0.3
    out = d_base[pick[0]] if alias else torch.full((n, 4), -1, dtype=torch.int64, device="cuda")
    ptrs = (C.c_void_p * max(count, 1))(*[d_base[j].data_ptr() for j in pick])
    cw = F.fr_encode(coeffs) if count else np.zeros((1, 4), dtype=np.uint64)
    _lib.check(lib.zkhip_fr_linear_combination_device(ptrs, cw.ctypes.data, count, n, out.data_ptr(), None))
    torch.cuda.synchronize()
    exp = [sum(c * base[j][i] for c, j in zip(coeffs, pick)) % R for i in range(n)]
    assert F.fr_decode(out.cpu().numpy().view(np.uint64)) == exp, f"linear combination n={n} count={count} alias={alias}"

def fuzz_rows_sum(seed):
    """zkhip_fr_eval_rows_sum_device: sum_p w_p program_p(row) for 1 .. 6 random programs over shared columns against the oracle's interpreter"""
    log_rows = rng.randint(0, 7); rows = 1 << log_rows; n_cols = rng.randint(1, 4)
    cols = [[rng.randrange(R) for _ in range(rows)] for _ in range(n_cols)]
    omega = O.omega_for(log_rows) if rng.random() < 0.5 else None
    progs, exp = [], [0] * rows
    weights = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(rng.randint(1, 6))]
    for wgt in weights:
        p = E.RowProgram(rot_scale=rng.choice([1, 2]), omega=omega)
        written = []
        def operand():
            kinds = ["const", "col"] + (["reg"] if written else []) + (["rowpow"] if omega else [])
            k = rng.choice(kinds)
            if k == "const": return p.constant(rng.choice([0, 1, R - 1, rng.randrange(R)]))
            if k == "col": return p.column(rng.randrange(n_cols), rng.choice([0, 1, -1, 3]))
            return E.RowProgram.reg(rng.choice(written)) if k == "reg" else E.RowProgram.ROWPOW
        n_regs = rng.choice([6, 8, 12, 16])
        for _ in range(rng.randint(1, 40)):
            dst = rng.randrange(n_regs); p.emit(rng.randrange(8), dst, operand(), operand(), operand())
            if dst not in written: written.append(dst)
        p.result_reg = rng.choice(written)
        progs.append(p)
        val = O.row_program_run(p.insns, p.constants, p.rotations, p.rot_scale, p.result_reg, cols + [cols[0]] * 8, log_rows, omega=omega)
        exp = [(e + wgt * v) % R for e, v in zip(exp, val)]
    d_cols = [torch.from_numpy(F.fr_encode(c).view(np.int64)).cuda() for c in cols]
    out = torch.full((rows, 4), -1, dtype=torch.int64, device="cuda")
    E.run_programs_sum_device(progs, weights, [t.data_ptr() for t in d_cols], log_rows, out.data_ptr())
    torch.cuda.synchronize()
    assert F.fr_decode(out.cpu().numpy().view(np.uint64)) == exp, f"sum of row programs rows={rows} programs={len(progs)}"

fns = [fuzz_msm, fuzz_host_chunked] if LARGE else [fuzz_msm, fuzz_ntt, fuzz_poly, fuzz_rows, fuzz_lookup, fuzz_batched, fuzz_sharded, fuzz_g2, fuzz_perm, fuzz_lincomb, fuzz_rows_sum]
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    f = fns[it % len(fns)]
    seed = rng.randrange(1 << 30)
    state = rng.getstate()
    try:
        f(seed)
    except AssertionError as e:
        print(f"MISMATCH in {f.__name__}: {e}  (run seed {seed0}, iteration {it}, case seed {seed})", flush=True)
        sys.exit(1)
    counts[f.__name__] = counts.get(f.__name__, 0) + 1
    it += 1
    if it % (5 if LARGE else 50) == 0: print(f"{it} cases ok {counts}", flush=True)
print(f"done: {it} cases, no mismatch {counts}")
