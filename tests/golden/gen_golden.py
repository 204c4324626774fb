#!/usr/bin/env python3
"""Generates tests/golden/*.json from oracle/bn254.py (the independent big-integer model).  The reference holds no
fixtures at this boundary, so these vectors are this repo's own: inputs + expected outputs in
the boundary's memory formats (hex of little-endian u64 Montgomery limbs)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import bn254 as O

HERE = os.path.dirname(os.path.abspath(__file__))
hx = lambda limbs: "".join("%016x" % w for w in limbs)   # 4 or 8 or 12 limbs, each 16 hex digits, limb 0 first


def msm_case(name, scalars, bases):
    exp = O.msm_naive(scalars, bases)
    assert exp == O.best_multiexp(scalars, bases, threads=3)
    return {"name": name, "scalars": [hx(O.fr_to_limbs(s)) for s in scalars], "bases": [hx(O.affine_to_limbs(P)) for P in bases],
            "expected_affine": hx(O.affine_to_limbs(exp))}


def main():
    g = O.SplitMix64(0x5A4B534E41500001)
    G = O.G1_GEN
    kat = {
        "two_G": hx(O.affine_to_limbs(O.scalar_mul(2, G))),
        "r_minus_1_G": hx(O.affine_to_limbs(O.scalar_mul(O.R_MOD - 1, G))),
        "k12345_G": hx(O.affine_to_limbs(O.scalar_mul(12345, G))),
        "fr_root_of_unity": hx(O.fr_to_limbs(O.FR_ROOT_OF_UNITY)),
        "fr_zeta": hx(O.fr_to_limbs(O.FR_ZETA)),
        "omega_22": "%064x" % O.omega_for(22),
        "omega_24": "%064x" % O.omega_for(24),
    }
    pts = [O.scalar_mul(g.fr(), G) for _ in range(48)]
    cases = []
    sc = [g.fr() for _ in range(48)]
    cases.append(msm_case("uniform48", sc, pts))
    sc2 = list(sc); sc2[0] = 0; sc2[1] = 1; sc2[2] = O.R_MOD - 1; sc2[3] = 1 << 253; sc2[4] = sc2[5]
    pts2 = list(pts); pts2[6] = None; pts2[7] = pts2[8]; pts2[9] = O.neg(pts2[10]); sc2[9] = sc2[10]
    cases.append(msm_case("edges48", sc2, pts2))
    cases.append(msm_case("all_zero_scalars", [0] * 8, pts[:8]))
    cases.append(msm_case("all_identity_bases", sc[:8], [None] * 8))
    cases.append(msm_case("single", sc[:1], pts[:1]))
    cases.append(msm_case("same_base_32", sc[:32], [pts[0]] * 32))
    s = g.fr()
    srs = O.structured_srs(s, 16)
    a = [g.fr() for _ in range(16)]
    c = msm_case("structured_srs16", a, srs)
    assert O.msm_naive(a, srs) == O.scalar_mul(sum(x * pow(s, i, O.R_MOD) for i, x in enumerate(a)) % O.R_MOD, G)
    cases.append(c)
    json.dump({"kat": kat, "msm": cases}, open(os.path.join(HERE, "msm_g1.json"), "w"), indent=0)

    ntt = []
    for L in (0, 1, 2, 3, 5, 7):
        a = [g.fr() for _ in range(1 << L)]
        w = O.omega_for(L)
        out = O.best_fft(a, w, L)
        assert out == O.dft_naive(a, w)
        ntt.append({"log_n": L, "omega": hx(O.fr_to_limbs(w)), "input": [hx(O.fr_to_limbs(x)) for x in a],
                    "expected": [hx(O.fr_to_limbs(x)) for x in out]})
    dom = O.EvaluationDomain(4, 4)
    a = [g.fr() for _ in range(dom.n)]
    ext = dom.coeff_to_extended(a)
    domain = {"j": 4, "k": 4, "coeffs": [hx(O.fr_to_limbs(x)) for x in a],
              "lagrange_to_coeff": [hx(O.fr_to_limbs(x)) for x in dom.lagrange_to_coeff(a)],
              "coeff_to_extended": [hx(O.fr_to_limbs(x)) for x in ext],
              "divide_by_vanishing_poly": [hx(O.fr_to_limbs(x)) for x in dom.divide_by_vanishing_poly(ext)],
              "extended_to_coeff": [hx(O.fr_to_limbs(x)) for x in dom.extended_to_coeff(ext)]}
    json.dump({"ntt": ntt, "domain": domain}, open(os.path.join(HERE, "ntt_fr.json"), "w"), indent=0)
    json.dump(prover_steps(g), open(os.path.join(HERE, "prover_steps.json"), "w"), indent=0)
    print("wrote msm_g1.json, ntt_fr.json, prover_steps.json")


class E:   # the three fields oracle.eval_expression reads of a plonk::Expression node
    def __init__(self, kind, a=None, b=None): self.kind, self.a, self.b = kind, a, b


class Obj:
    def __init__(self, **kw): self.__dict__.update(kw)


def halo2_lib_like_cs(gate_cols, blinding):
    """the constraint-system shape of tests/test_gpu_rows.py::halo2_lib_like_cs, as plain oracle-side objects"""
    A = gate_cols
    adv, fix = (lambda c, r=0: E("advice", c, r)), (lambda c, r=0: E("fixed", c, r))
    sub = lambda x, y: E("sum", x, E("neg", y))
    gates = [[E("product", fix(i), sub(E("sum", adv(i, 0), E("product", adv(i, 1), adv(i, 2))), adv(i, 3)))] for i in range(A)]
    lookups = [Obj(input_expressions=[adv(A)], table_expressions=[fix(A)]),
               Obj(input_expressions=[E("product", adv(0), fix(A + 1)), E("sum", adv(1, -1), E("constant", 5))],
                   table_expressions=[fix(A), E("scaled", fix(A + 1), 3)])]
    perm = [("advice", i) for i in range(A)] + [("fixed", A + 1), ("instance", 0)]
    return Obj(num_fixed=A + 2, num_advice=A + 1, num_instance=1, gates=gates, lookups=lookups, permutation_columns=perm,
               blinding_factors=blinding, degree=4)


def prover_steps(g):
    fr = lambda v: hx(O.fr_to_limbs(v))
    # 1. a row program in the ABI's own terms (include/zkhip.h): every opcode and operand kind, rotations with wrap-around, PREV
    log_rows, rot_scale = 4, 2
    rows = 1 << log_rows
    cols = [[g.fr() for _ in range(rows)] for _ in range(3)]
    cols[0][0], cols[0][1], cols[1][0] = 0, O.R_MOD - 1, O.R_MOD - 1
    prev = [g.fr() for _ in range(rows)]
    consts, rots, omega = [g.fr(), 1, O.R_MOD - 1, 0], [0, 1, -1, 3], O.omega_for(log_rows)
    C_, R_, COL, PREV, POW = (lambda i: (0, i, 0)), (lambda i: (1, i, 0)), (lambda c, r: (2, c, r)), (3, 0, 0), (4, 0, 0)
    z = (0, 0, 0)
    insns = [(0, 0, COL(0, 0), z, z), (1, 1, R_(0), COL(1, 1), z), (2, 2, R_(1), COL(2, 2), z), (3, 3, R_(2), R_(1), z),
             (4, 4, R_(3), z, z), (5, 5, R_(4), z, z), (6, 6, R_(5), z, z), (7, 7, R_(6), C_(0), PREV),
             (3, 8, POW, COL(0, 3), z), (7, 0, R_(8), R_(7), C_(2)), (2, 9, C_(3), R_(0), z), (1, 15, R_(9), POW, z)]
    row_program = {"log_rows": log_rows, "rot_scale": rot_scale, "rotations": rots, "constants": [fr(c) for c in consts], "omega": fr(omega),
                   "result_reg": 15, "insns": [[op, dst] + list(a) + list(b) + list(c) for op, dst, a, b, c in insns],
                   "columns": [[fr(x) for x in c] for c in cols], "prev": [fr(x) for x in prev],
                   "expected": [fr(x) for x in O.row_program_run(insns, consts, rots, rot_scale, 15, cols, log_rows, omega=omega, prev=prev)]}
    # 2. evaluate_h on the halo2-lib shape, from the direct restatement of the formulas
    k, ek, gate_cols, blinding = 3, 5, 2, 2
    cs = halo2_lib_like_cs(gate_cols, blinding)
    n_perm, sets = len(cs.permutation_columns), (len(cs.permutation_columns) + 1) // 2
    total = cs.num_fixed + cs.num_advice + cs.num_instance + 3 + n_perm + sets + 3 * len(cs.lookups)
    c = [[g.fr() for _ in range(1 << ek)] for _ in range(total)]
    beta, gamma, theta, y = g.fr(), g.fr(), g.fr(), g.fr()
    o = 0
    fixed = c[o:o + cs.num_fixed]; o += cs.num_fixed
    advice = c[o:o + cs.num_advice]; o += cs.num_advice
    inst = c[o:o + 1]; o += 1
    l0, l_last, l_active = c[o], c[o + 1], c[o + 2]; o += 3
    sigma = c[o:o + n_perm]; o += n_perm
    zs = c[o:o + sets]; o += sets
    lk = [tuple(c[o + 3 * i + j] for j in range(3)) for i in range(len(cs.lookups))]
    exp = O.evaluate_h_direct(cs, k, ek, fixed, advice, inst, l0, l_last, l_active, sigma, zs, lk, beta, gamma, theta, y)
    evaluate_h = {"k": k, "extended_k": ek, "gate_cols": gate_cols, "blinding_factors": blinding, "beta": fr(beta), "gamma": fr(gamma),
                  "theta": fr(theta), "y": fr(y), "columns": [[fr(x) for x in col] for col in c], "expected": [fr(x) for x in exp]}
    # 3. lookup permutation and grand product
    usable = 26
    table = [i % 8 for i in range(32)]
    inputs = [table[g.next() % usable] for _ in range(32)]
    pi, pt = O.permute_expression_pair(inputs, table, usable)
    num, den = [g.fr() for _ in range(32)], [g.fr() for _ in range(32)]
    den[9] = 0
    return {"row_program": row_program, "evaluate_h": evaluate_h,
            "lookup_permute": {"usable_rows": usable, "input": [fr(x) for x in inputs], "table": [fr(x) for x in table],
                               "permuted_input": [fr(x) for x in pi], "permuted_table": [fr(x) for x in pt]},
            "grand_product": {"num": [fr(x) for x in num], "den": [fr(x) for x in den], "z": [fr(x) for x in O.grand_product(num, den)]}}


if __name__ == "__main__":
    main()
