"""CPU: the host logic of the multi-open provers (zksnap_circuits_halo2_amd/multiopen.py) -- query grouping of GWC, rotation sets of
SHPLONK in the order the reference builds them (polynomials and sets by first appearance, points ascending as a BTreeSet iterates),
the low-degree interpolation and the vanishing-polynomial evaluations -- against the oracle's polynomial arithmetic.  The device side of
the same provers is covered by tests/test_gpu_multiopen.py."""
import random

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import multiopen as M

R = O.R_MOD


def test_gwc_groups_queries_by_point_in_order_of_first_appearance():
    x, y, z = 11, 5, 7
    qs = [M.ProverQuery(x, 100), M.ProverQuery(y, 101), M.ProverQuery(x, 102), M.ProverQuery(z, 100), M.ProverQuery(y, 103)]
    sets = M.construct_intermediate_sets(qs)
    assert [p for p, _ in sets] == [x, y, z]
    assert [[q.poly for q in grp] for _, grp in sets] == [[100, 102], [101, 103], [100]]
    assert sets[0][1][0] is qs[0] and sets[1][1][1] is qs[4]                     # the queries themselves, not copies (evaluations are filled in place)


def test_shplonk_rotation_sets_follow_the_reference_order():
    rng = random.Random(3)
    a, b, c = sorted(rng.randrange(R) for _ in range(3))
    plan = [(7, c), (7, a), (8, b), (9, a), (9, c), (10, b), (11, a), (11, b), (11, c), (12, c), (12, a)]
    qs = [M.ProverQuery(pt, poly, eval=(poly * 1000 + pt) % R) for poly, pt in plan]
    sets, super_points = M.construct_rotation_sets(qs)
    assert super_points == [a, b, c]
    # sets in order of the first polynomial that has them; points ascending; polynomials in order of first appearance
    assert [rs.points for rs in sets] == [[a, c], [b], [a, b, c]]
    assert [rs.polys for rs in sets] == [[7, 9, 12], [8, 10], [11]]
    for rs in sets:                                                              # evals[j][i] = evaluation of polys[j] at points[i]
        for poly, ev in zip(rs.polys, rs.evals):
            assert ev == [(poly * 1000 + pt) % R for pt in rs.points]


def test_interpolation_and_vanishing_evaluations_against_the_oracle():
    rng = random.Random(4)
    for m in (1, 2, 3, 5):
        pts = [rng.randrange(R) for _ in range(m)]
        coeffs = [rng.randrange(R) for _ in range(m)]                            # a polynomial of degree < m ...
        evals = [O.eval_polynomial(coeffs, p) for p in pts]
        assert M._interpolate(pts, evals) == coeffs                              # ... is recovered from its m evaluations
        x = rng.randrange(R)
        assert M._eval_small(coeffs, x) == O.eval_polynomial(coeffs, x)
        v = 1
        for p in pts:
            v = v * (x - p) % R
        assert M._vanishing_at(pts, x) == v
    assert M._vanishing_at([], 12345) == 1
