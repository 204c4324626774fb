"""CPU: pins the oracle (oracle/bn254.py, oracle/cpu_ref.c) to public constants, known answers, the committed golden
vectors and to each other.  The reference itself holds no vectors at this boundary (parity unpinned)."""
import json
import os

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import fields as F

GOLD = os.path.join(os.path.dirname(__file__), "golden")
unhex = lambda s: [int(s[i:i + 16], 16) for i in range(0, len(s), 16)]


def test_field_constants_match_survey():
    # Montgomery constants of halo2curves bn256::{Fr,Fq} [DEP]
    assert O.limbs4(O.R_MOD) == [0x43e1f593f0000001, 0x2833e84879b97091, 0xb85045b68181585d, 0x30644e72e131a029]
    assert O.limbs4(O.MONT_R % O.R_MOD) == [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f]
    assert O.limbs4(O.MONT_R ** 2 % O.R_MOD) == [0x1bb8e645ae216da7, 0x53fe3ab1e35c59e3, 0x8c49833d53bb8085, 0x0216d0b17f4e44a5]
    assert O.mont_inv64(O.R_MOD) == 0xc2e1f593efffffff
    assert O.limbs4(O.Q_MOD) == [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029]
    assert O.limbs4(O.MONT_R % O.Q_MOD) == [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
    assert O.limbs4(O.MONT_R ** 2 % O.Q_MOD) == [0xf32cfc5b538afa89, 0xb5e71911d44501fb, 0x47ab1eff0a417ff6, 0x06d89f71cab8351f]
    assert O.mont_inv64(O.Q_MOD) == 0x87d20782e4866389


def test_roots_of_unity():
    assert O.FR_ROOT_OF_UNITY == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert pow(O.FR_ROOT_OF_UNITY, 1 << 28, O.R_MOD) == 1 and pow(O.FR_ROOT_OF_UNITY, 1 << 27, O.R_MOD) != 1
    assert O.omega_for(22) == 0x18c95f1ae6514e11a1b30fd7923947c5ffcec5347f16e91b4dd654168326bede
    assert O.omega_for(24) == 0x1951441010b2b95a6e47a6075066a50a036f5ba978c050f2821df86636c0facb
    assert pow(O.FR_ZETA, 3, O.R_MOD) == 1 and O.FR_ZETA != 1


def test_curve_known_answers():
    G = O.G1_GEN
    assert O.on_curve(G)
    # EIP-196 ecMul vector
    assert O.scalar_mul(2, G) == (0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3,
                                  0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)
    assert O.scalar_mul(O.R_MOD, G) is None
    assert O.scalar_mul(O.R_MOD - 1, G) == (1, O.Q_MOD - 2)
    assert O.scalar_mul(12345, G) == (0x1936f7b07be20ac4b7faac53aba252c44112b369f437c12d75b8157882b390aa,
                                      0x055c38c27b1dc7fbbdfbb7b4795e92d0d838126c25b6771908f9a23c35c8921a)
    assert O.add(G, O.neg(G)) is None and O.add(None, G) == G


def test_python_oracle_identities():
    g = O.SplitMix64(42)
    s = g.fr()
    srs = O.structured_srs(s, 24)
    a = [g.fr() for _ in range(24)]
    horner = sum(x * pow(s, i, O.R_MOD) for i, x in enumerate(a)) % O.R_MOD
    assert O.best_multiexp(a, srs, threads=1) == O.best_multiexp(a, srs, threads=5) == O.scalar_mul(horner, O.G1_GEN)
    for L in (0, 1, 4, 6):
        v = [g.fr() for _ in range(1 << L)]
        w = O.omega_for(L)
        f = O.best_fft(v, w, L)
        assert f == O.dft_naive(v, w)
        ninv = pow(1 << L, -1, O.R_MOD)
        assert [x * ninv % O.R_MOD for x in O.best_fft(f, pow(w, -1, O.R_MOD), L)] == v
    assert O.best_fft([1] + [0] * 15, O.omega_for(4), 4) == [1] * 16
    d = O.EvaluationDomain(4, 5)
    p = [g.fr() for _ in range(d.n)]
    assert d.extended_k == 7 and len(d.t_evaluations) == 4
    assert d.extended_to_coeff(d.coeff_to_extended(p)) == p + [0] * (2 * d.n)
    assert d.coeff_to_lagrange(d.lagrange_to_coeff(p)) == p


def test_golden_msm_vs_both_oracles(cref):
    data = json.load(open(os.path.join(GOLD, "msm_g1.json")))
    assert unhex(data["kat"]["two_G"]) == O.affine_to_limbs(O.scalar_mul(2, O.G1_GEN))
    for case in data["msm"]:
        sc = np.array([unhex(s) for s in case["scalars"]], dtype=np.uint64).reshape(-1, 4)
        bs = np.array([unhex(s) for s in case["bases"]], dtype=np.uint64).reshape(-1, 8)
        exp = unhex(case["expected_affine"])
        ints = [O.fr_from_limbs([int(x) for x in r]) for r in sc]
        pts = [O.affine_from_limbs([int(x) for x in r]) for r in bs]
        assert O.affine_to_limbs(O.msm_naive(ints, pts)) == exp, case["name"]
        for threads in (1, 3):
            got = cref.jac_to_affine(cref.best_multiexp(sc, bs, threads))
            assert [int(x) for x in got] == exp, (case["name"], threads)


def test_golden_ntt_vs_both_oracles(cref):
    data = json.load(open(os.path.join(GOLD, "ntt_fr.json")))
    for case in data["ntt"]:
        L = case["log_n"]
        a = np.array([unhex(s) for s in case["input"]], dtype=np.uint64).reshape(-1, 4)
        exp = np.array([unhex(s) for s in case["expected"]], dtype=np.uint64).reshape(-1, 4)
        omega = np.array(unhex(case["omega"]), dtype=np.uint64)
        assert F.fr_encode(O.best_fft(F.fr_decode(a), F.fr_decode(omega)[0], L)).tolist() == exp.tolist()
        for threads in (1, 4):
            b = a.copy()
            cref.best_fft(b, omega, L, threads)
            assert np.array_equal(b, exp), (L, threads)


@pytest.mark.parametrize("n,threads", [(1, 1), (2, 2), (33, 1), (100, 7), (1000, 8)])
def test_c_oracle_msm_vs_structured_identity(cref, n, threads):
    bases, t0, d = cref.gen_bases(1000 + n, n)
    sc = cref.gen_scalars(2000 + n, n, n % 2)
    got = cref.jac_to_affine(cref.best_multiexp(sc, bases, threads))
    exp = cref.jac_to_affine(cref.scalar_mul(cref.expected_scalar(sc, t0, d), cref.generator()))
    assert np.array_equal(got, exp)


def test_c_oracle_matches_python_generators_and_fields(cref):
    g = O.SplitMix64(77)
    assert F.fr_decode(cref.gen_scalars(77, 50, 0)) == [g.fr() for _ in range(50)]
    assert F.fr_decode(cref.gen_scalars(99, 300, 1)) == O.witness_like_scalars(99, 300)
    bases, t0, d = cref.gen_bases(11, 12)
    assert [O.affine_from_limbs([int(x) for x in r]) for r in bases] == [O.scalar_mul((t0 + i * d) % O.R_MOD, O.G1_GEN) for i in range(12)]
    for field, p in ((0, O.Q_MOD), (1, O.R_MOD)):
        a = [g.fr() % p for _ in range(64)] + [0, 1, p - 1]
        b = [g.fr() % p for _ in range(64)] + [p - 1, p - 1, p - 1]
        enc = lambda v: np.array([O.limbs4(O.to_mont(x, p)) for x in v], dtype=np.uint64)
        for op, f in ((0, lambda x, y: x * y % p), (1, lambda x, y: (x + y) % p), (2, lambda x, y: (x - y) % p), (3, lambda x, y: x * x % p)):
            assert np.array_equal(cref.field_op(field, op, enc(a), enc(b)), enc([f(x, y) for x, y in zip(a, b)]))


def test_c_oracle_domain_passes(cref):
    d = O.EvaluationDomain(4, 6)
    g = O.SplitMix64(5)
    p = [g.fr() for _ in range(d.n)]
    a = F.fr_encode(p)
    enc1 = lambda v: F.fr_encode([v])[0]
    # coeff_to_extended = distribute_powers_zeta + pad + fft
    ext = np.zeros((d.extended_len(), 4), dtype=np.uint64)
    ext[: d.n] = a
    cref.distribute_powers_zeta(ext[: d.n], enc1(d.g_coset), enc1(d.g_coset_inv))
    cref.best_fft(ext, enc1(d.extended_omega), d.extended_k, 2)
    assert F.fr_decode(ext) == d.coeff_to_extended(p)
    q = ext.copy()
    cref.mul_periodic(q, F.fr_encode(d.t_evaluations))
    assert F.fr_decode(q) == d.divide_by_vanishing_poly(F.fr_decode(ext))
    cref.best_fft(ext, enc1(d.extended_omega_inv), d.extended_k, 2)
    cref.scale(ext, enc1(d.extended_ifft_divisor))
    cref.distribute_powers_zeta(ext, enc1(d.g_coset_inv), enc1(d.g_coset))
    assert F.fr_decode(ext[: 3 * d.n]) == p + [0] * (2 * d.n)


def test_row_a7_oracles_agree(cref):
    """eval_polynomial / kate_division / batch_invert / prefix_product: C restatement == big-int model, plus identities."""
    g = O.SplitMix64(31)
    for n in (1, 2, 5, 33, 200):
        a = [g.fr() for _ in range(n)]
        x = g.fr()
        A, X = F.fr_encode(a), F.fr_encode([x])[0]
        assert F.fr_decode(cref.eval_polynomial(A, X))[0] == O.eval_polynomial(a, x) == sum(c * pow(x, i, O.R_MOD) for i, c in enumerate(a)) % O.R_MOD
        q = O.kate_division(a, x)
        assert F.fr_decode(cref.kate_division(A, X)) == q
        if n > 1:   # a(X) = q(X) (X - x) + a(x): check at a random point
            y = g.fr()
            assert (O.eval_polynomial(q, y) * (y - x) + O.eval_polynomial(a, x)) % O.R_MOD == O.eval_polynomial(a, y)
        v = list(a)
        if n > 3:
            v[1] = 0
        V = F.fr_encode(v)
        assert F.fr_decode(cref.prefix_product(V)) == O.prefix_product(v)
        W = V.copy()
        cref.batch_invert(W)
        inv = F.fr_decode(W)
        assert inv == O.batch_invert(v) and all((p * q_) % O.R_MOD == (1 if p else 0) for p, q_ in zip(v, inv))


# ---------------------------------------------------------------- G2 (Fq2, twist curve): pinned to public values
def test_g2_oracle_public_facts():
    """the alt_bn128 G2 generator of EIP-197 lies on the twist y^2 = x^3 + 3/(9+u), has order r, and the Jacobian formulas agree with
    an independent affine implementation (zksnap_circuits_halo2_amd/srs.py, written for ParamsKZG::setup's s_g2)"""
    from zksnap_circuits_halo2_amd import srs

    assert O.g2_on_curve(O.G2_GEN)
    assert O.f2_mul(O.G2_B, (9, 1)) == (3, 0)
    assert O.g2_scalar_mul(O.R_MOD - 1, O.G2_GEN) == O.g2_neg(O.G2_GEN)
    assert O.g2_add(O.g2_scalar_mul(O.R_MOD - 1, O.G2_GEN), O.G2_GEN) is None
    assert O.f2_mul((0, 1), (0, 1)) == (O.Q_MOD - 1, 0)                      # u^2 = -1
    g = O.SplitMix64(222)
    for _ in range(4):
        k = g.fr()
        P = O.g2_scalar_mul(k, O.G2_GEN)
        assert O.g2_on_curve(P) and P == srs.g2_mul(k)
        assert O.g2_add(P, P) == O.g2_scalar_mul(2 * k, O.G2_GEN)
    a, b = g.fr(), g.fr()
    lhs = O.g2_msm_naive([a, b, 0], [O.G2_GEN, O.g2_scalar_mul(5, O.G2_GEN), O.G2_GEN])
    assert lhs == O.g2_scalar_mul((a + 5 * b) % O.R_MOD, O.G2_GEN)
    enc = O.g2_affine_to_limbs(O.G2_GEN)
    one = O.limbs4(O.to_mont(1, O.Q_MOD))
    assert O.g2_jac_from_limbs(enc + one + [0, 0, 0, 0]) == O.G2_GEN
    assert enc == [int(x) for x in srs.g2_encode(srs.G2_GENERATOR)]


def test_the_twist_has_the_same_endomorphism_with_beta_squared():
    """csrc/msm_g2.hip (round 5) reuses the G1 path's GLV split: on the twist y^2 = x^3 + 3 / (9 + u) the map (x, y) -> (beta' x, y) multiplies
    points of the order-r subgroup by the SAME lambda as on G1 exactly when beta' = beta^2 (beta: the constant of msm.hip's GLV).  Checked on
    the EIP-197 generator and on a second point; the kernel source carries beta^2 in Montgomery words."""
    import os
    import re

    r, q = O.R_MOD, O.Q_MOD
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
    beta2 = beta * beta % q
    assert pow(beta2, 3, q) == 1 and beta2 != 1
    phi = lambda P, b: ((P[0][0] * b % q, P[0][1] * b % q), P[1])
    for P in (O.G2_GEN, O.g2_scalar_mul(0xC0FFEE, O.G2_GEN)):
        assert O.g2_on_curve(phi(P, beta2))
        assert O.g2_scalar_mul(lam, P) == phi(P, beta2)
        assert O.g2_scalar_mul(lam, P) != phi(P, beta)               # beta itself acts as lambda^2 there
        assert O.g2_scalar_mul(lam * lam % r, P) == phi(P, beta)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zksnap_circuits_halo2_amd", "csrc", "msm_g2.hip")).read()
    m = re.search(r"constexpr uint32_t BETA2_EXT\[8\] = \{([^}]*)\}", src)
    assert [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")] == [(beta2 * (1 << 256) % q >> (32 * i)) & 0xffffffff for i in range(8)]


def test_glv_parameters_of_the_general_path_msm():
    """csrc/msm.hip splits the scalars of the general-path MSM with the curve's endomorphism phi(x, y) = (beta x, y) = lambda (x, y).  The constants
    in the kernel source are re-derived here: lambda / beta are matching cube roots of unity (checked on the generator), the lattice basis comes out
    of the extended Euclid on (r, lambda), g1 / g2 are its rounded quotients, and the word-level decomposition the kernel performs (floors,
    differences mod 2^160) stays below 9/8 (a1 + a2) < 0.979 * 2^127 -- so the signed recoding never carries out of the top window."""
    import math
    import os
    import random
    import re

    r, q = O.R_MOD, O.Q_MOD
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    beta = 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe
    assert pow(lam, 3, r) == 1 and lam != 1 and pow(beta, 3, q) == 1 and beta != 1
    G = O.G1_GEN
    assert O.scalar_mul(lam, G) == (beta * G[0] % q, G[1])
    P = O.scalar_mul(0xC0FFEE, G)
    assert O.scalar_mul(lam, P) == (beta * P[0] % q, P[1])
    # extended Euclid on (r, lambda): the two short vectors (a, b) with a + b lambda = 0 mod r
    rows = [(r, 1, 0), (lam, 0, 1)]
    while rows[-1][0] != 0:
        (r0, s0, t0), (r1, s1, t1) = rows[-2], rows[-1]
        qq = r0 // r1
        rows.append((r0 - qq * r1, s0 - qq * s1, t0 - qq * t1))
    sq = math.isqrt(r)
    li = next(i for i in range(len(rows) - 1) if rows[i][0] >= sq and rows[i + 1][0] < sq)
    a1, b1 = rows[li + 1][0], -rows[li + 1][2]
    a2, b2 = min(((rows[li][0], -rows[li][2]), (rows[li + 2][0], -rows[li + 2][2])), key=lambda v: v[0] ** 2 + v[1] ** 2)
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    assert (a1, -b1, a2, b2) == (0x89d3256894d213e3, 0x6f4d8248eeb859fc8211bbeb7d4f1128, 0x6f4d8248eeb859fd0be4e1541221250b, 0x89d3256894d213e3)
    rdiv = lambda a, n: (2 * a + n) // (2 * n)
    g1, g2 = rdiv(b2 << 256, r), rdiv((-b1) << 256, r)
    words = lambda v, n: [(v >> (32 * i)) & 0xffffffff for i in range(n)]
    # the kernel source carries exactly these words
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zksnap_circuits_halo2_amd", "csrc", "msm.hip")).read()
    def in_source(name, vals):
        m = re.search(r"constexpr uint32_t %s\[\d+\] = \{([^}]*)\}" % name, src)
        assert m, name
        assert [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")] == vals, name
    in_source("G1", words(g1, 3)); in_source("G2", words(g2, 5)); in_source("A1", words(a1, 2)); in_source("A2", words(a2, 4)); in_source("NB1", words(-b1, 4))
    in_source("BETA_EXT", words(beta * (1 << 256) % q, 8))
    M = (1 << 160) - 1
    rnd = random.Random(99)
    worst = 0
    for k in [0, 1, 2, r - 1, r - 2, lam, r - lam, r // 2, r // 3] + [rnd.randrange(r) for _ in range(20000)] + [rnd.randrange(1 << rnd.randrange(1, 254)) for _ in range(5000)]:
        c1, c2 = (k * g1) >> 256, (k * g2) >> 256
        assert c1 < 1 << 64 and c2 < 1 << 128
        k1 = ((k & M) - ((c1 * a1) & M) - ((c2 * a2) & M)) & M
        k2 = (((c1 * -b1) & M) - ((c2 * b2) & M)) & M
        v = []
        for x in (k1, k2):
            neg = x >> 159
            mag = (-x) & M if neg else x
            worst = max(worst, mag)
            v.append(-mag if neg else mag)
        assert (v[0] + v[1] * lam - k) % r == 0
    # floors of the APPROXIMATE quotients: c = floor(x + e) with |e| <= k / 2^257 < 1/8, so |k_i| < (1 + 1/8) (a1 + a2) -- which must leave room for the
    # signed recoding's carry in the top window when the windows cover exactly 128 bits (8-bit windows: top value <= 125, + 1 < 128)
    assert worst < (a1 + a2) * 9 // 8 + (1 << 64) < (1 << 127) - (1 << 120)


@pytest.mark.parametrize("n,bits,seed", [(1, 3, 1), (64, 3, 2), (1000, 6, 3), (5000, 10, 4), (1 << 14, 12, 5), (3000, 40, 6)])
def test_numpy_permute_expression_pair_is_the_statement_by_statement_one(n, bits, seed):
    """oracle.permute_expression_pair_np (the vectorised form the 2^22-row GPU test uses) against the statement-by-statement restatement of
    `permute_expression_pair` [DEP halo2-axiom plonk/lookup/prover.rs], incl. the missing-value error"""
    import random

    rng = random.Random(seed)
    usable = max(1, n - 6) if n > 8 else n
    table = [i % (1 << bits) for i in range(n)] if bits < 20 else [rng.randrange(1 << bits) for _ in range(n)]
    inputs = [rng.choice(table[:usable]) for _ in range(n)]
    head = table[:usable]
    rng.shuffle(head)
    table[:usable] = head
    exp_in, exp_tab = O.permute_expression_pair(inputs, table, usable)
    got_in, got_tab = O.permute_expression_pair_np(inputs, table, usable)
    assert got_in.tolist() == exp_in and got_tab.tolist() == exp_tab
    if n > 8:
        bad = list(inputs)
        bad[3] = (1 << 62) + 5
        with pytest.raises(ValueError):
            O.permute_expression_pair_np(bad, table, usable)
        with pytest.raises(ValueError):
            O.permute_expression_pair(bad, table, usable)
