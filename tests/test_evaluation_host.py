"""CPU (-m "not gpu"): host logic of the `plonk/evaluation.rs` mirror -- expression graph, hash-consing, register allocation
and the evaluate_h program builder -- checked without a GPU by interpreting the compiled program with the oracle's big-int
interpreter and comparing it with the oracle's direct restatement of the evaluate_h formulas."""
import ctypes as C
import random

import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, evaluation as E

R = O.R_MOD


def make_cs(A):
    gates = [[E.Fixed(i) * (E.Advice(i, 0) + E.Advice(i, 1) * E.Advice(i, 2) - E.Advice(i, 3))] for i in range(A)]
    lookups = [E.Lookup([E.Advice(A)], [E.Fixed(A)]),
               E.Lookup([E.Advice(0) * E.Fixed(A + 1), E.Advice(1, -1) + E.Constant(5)], [E.Fixed(A), E.Fixed(A + 1) * 3])]
    perm = [("advice", i) for i in range(A)] + [("fixed", A + 1), ("instance", 0)]
    return E.ConstraintSystem(num_fixed=A + 2, num_advice=A + 1, num_instance=1, gates=gates, lookups=lookups,
                              permutation_columns=perm, blinding_factors=3, degree=4)


@pytest.mark.parametrize("A,k,ek", [(1, 2, 4), (3, 3, 5), (5, 3, 5)])
def test_compiled_evaluate_h_equals_direct_formulas(A, k, ek):
    rng = random.Random(A)
    cs = make_cs(A)
    qc = E.quotient_columns(cs)
    rows = 1 << ek
    cols = [[rng.randrange(R) for _ in range(rows)] for _ in range(qc.total)]
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prog = E.evaluate_h_program(cs, k, ek, beta, gamma, theta, y)
    assert prog.n_columns == qc.total and prog.rot_scale == 1 << (ek - k) and prog.omega == O.omega_for(ek)
    assert max(i[1] for i in prog.insns) < _lib.VM_REGS
    got = O.row_program_run(prog.insns, prog.constants, prog.rotations, prog.rot_scale, prog.result_reg, cols, ek, omega=prog.omega)
    exp = O.evaluate_h_direct(
        cs, k, ek, cols[qc.fixed:qc.fixed + cs.num_fixed], cols[qc.advice:qc.advice + cs.num_advice],
        cols[qc.instance:qc.instance + 1], cols[qc.l0], cols[qc.l_last], cols[qc.l_active_row],
        cols[qc.sigma:qc.sigma + len(cs.permutation_columns)], cols[qc.perm_product:qc.perm_product + cs.num_permutation_sets],
        [tuple(cols[qc.lookup + 3 * i + j] for j in range(3)) for i in range(2)], beta, gamma, theta, y)
    assert got == exp


def test_graph_hash_consing_and_register_reuse():
    g = E.Graph()
    a, b = g.col(0), g.col(1, 2)
    s1, s2 = g.add(a, b), g.add(b, a)
    assert s1 == s2 and g.col(0) == a and g.const(5) == g.const(5 + R)
    assert g.mul(s1, s1) == g.op(E.OP_SQR, s1)
    # a long dependent chain needs two registers at most, whatever its length
    v = g.mul(a, b)
    for i in range(100):
        v = g.add(g.mul(v, g.const(i + 2)), a)
    prog = E.compile_graph(g, v)
    assert len(prog.insns) == 201 and max(i[1] for i in prog.insns) <= 1
    cols = [[3, 4], [5, 6]]
    exp = []
    for r in range(2):
        x = cols[0][r] * cols[1][(r + 2) % 2]
        for i in range(100):
            x = (x * (i + 2) + cols[0][r]) % R
        exp.append(x)
    assert O.row_program_run(prog.insns, prog.constants, prog.rotations, 1, prog.result_reg, cols, 1) == exp


def test_register_pressure_is_reported():
    g = E.Graph()
    leaves = [g.mul(g.col(i), g.col(i + 1)) for i in range(_lib.VM_REGS + 1)]     # all live until the final sum
    acc = None
    for x in reversed(leaves):
        acc = x if acc is None else g.add(x, acc)
    # creation order computes every product first: one register too many
    with pytest.raises(ValueError, match="live registers"):
        E.compile_graph(g, acc)


def test_leaf_result_and_marshalling_layout():
    g = E.Graph()
    prog = E.compile_graph(g, g.col(2, -1))
    assert prog.insns == [(E.OP_MOV, 0, (E.SRC_COLUMN, 2, 0), (0, 0, 0), (0, 0, 0))] and prog.rotations == [-1]
    p, keep = prog._marshal()
    assert C.sizeof(_lib.VmInsn) == 16 and p.n_insns == 1 and p.n_rotations == 1 and p.n_constants == 0
    raw = bytes(C.string_at(C.addressof(p.insns[0]), 16))
    assert raw == bytes([E.OP_MOV, 0, 0, 0, E.SRC_COLUMN, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0])   # op dst rsvd | kind rot index(LE) x3


def _unhex(s):
    return sum(int(s[16 * i:16 * i + 16], 16) << (64 * i) for i in range(4)) * pow(1 << 256, -1, R) % R


def test_golden_prover_steps_against_oracle_and_compiler():
    """tests/golden/prover_steps.json (generated from the oracle alone): the oracle reproduces it, and the host compiler's program
    for the same constraint system, interpreted by the oracle, gives the fixture's evaluate_h column"""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prover_steps.json")))
    rp = gold["row_program"]
    insns = [(r[0], r[1], tuple(r[2:5]), tuple(r[5:8]), tuple(r[8:11])) for r in rp["insns"]]
    cols = [[_unhex(x) for x in c] for c in rp["columns"]]
    got = O.row_program_run(insns, [_unhex(x) for x in rp["constants"]], rp["rotations"], rp["rot_scale"], rp["result_reg"], cols,
                            rp["log_rows"], omega=_unhex(rp["omega"]), prev=[_unhex(x) for x in rp["prev"]])
    assert got == [_unhex(x) for x in rp["expected"]]
    eh = gold["evaluate_h"]
    cs = make_cs(eh["gate_cols"])
    cs.blinding_factors = eh["blinding_factors"]
    prog = E.evaluate_h_program(cs, eh["k"], eh["extended_k"], *(_unhex(eh[c]) for c in ("beta", "gamma", "theta", "y")))
    cols = [[_unhex(x) for x in c] for c in eh["columns"]]
    got = O.row_program_run(prog.insns, prog.constants, prog.rotations, prog.rot_scale, prog.result_reg, cols, eh["extended_k"], omega=prog.omega)
    assert got == [_unhex(x) for x in eh["expected"]]
    lp = gold["lookup_permute"]
    pi, pt = O.permute_expression_pair([_unhex(x) for x in lp["input"]], [_unhex(x) for x in lp["table"]], lp["usable_rows"])
    assert pi == [_unhex(x) for x in lp["permuted_input"]] and pt == [_unhex(x) for x in lp["permuted_table"]]
    gp = gold["grand_product"]
    assert O.grand_product([_unhex(x) for x in gp["num"]], [_unhex(x) for x in gp["den"]]) == [_unhex(x) for x in gp["z"]]
