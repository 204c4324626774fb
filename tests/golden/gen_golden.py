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
    print("wrote msm_g1.json, ntt_fr.json")


if __name__ == "__main__":
    main()
