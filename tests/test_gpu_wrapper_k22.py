"""GPU, BASELINE configs[3] at its real size: the wrapper circuit's non-MSM / non-NTT phases at k = 22
(/root/reference/aggregator/benches/wrapper_circuit.rs:21 `k = 22`, :61-68 the BaseCircuitParams: 4 advice, 1 lookup column, lookup_bits = 21).

Until round 4 these sizes ran only inside bench.py (as property checks whose `false` would not fail the line); here they are parity tests:
  * the whole device-resident prover flow at k = 22 (tools/prove_flow.py: all five invariants, and the corrupted-gate negative),
  * `evaluate_h` over the 2^24-row extended coset of the halo2-lib shape, ~150 rows (incl. the wrap-around rows) against the oracle's row interpreter,
  * the permutation / lookup grand product at 2^22 against the C oracle (batch inversion, product, running product),
  * `permute_expression_pair` at 2^22 usable rows of a 21-bit range lookup against the oracle's restatement."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F

pytestmark = pytest.mark.gpu
R = O.R_MOD
K, EK = 22, 24


def to_montgomery(lib, plain_words: np.ndarray) -> np.ndarray:
    """(n, 4) plain little-endian integers -> Montgomery words, through the library's field multiply (x * R^2 / R)"""
    r2 = np.ascontiguousarray(np.broadcast_to(F.fr_encode([pow(2, 256, R)])[0], plain_words.shape))
    out = np.empty_like(plain_words)
    _lib.check(lib.zkhip_test_field_op(1, 0, plain_words.ctypes.data, r2.ctypes.data, out.ctypes.data, plain_words.shape[0]))
    return out


def from_montgomery(lib, words: np.ndarray) -> np.ndarray:
    one = np.zeros_like(words)
    one[:, 0] = 1
    out = np.empty_like(words)
    _lib.check(lib.zkhip_test_field_op(1, 0, np.ascontiguousarray(words).ctypes.data, one.ctypes.data, out.ctypes.data, words.shape[0]))
    return out


def test_prover_flow_at_the_wrapper_size(lib):
    """configs[3]: SRS, advice commitments, permutation + lookup arguments, 2^22 iNTTs, 2^24 coset NTTs, the fused quotient over 2^24 rows,
    h commitments, evaluations -- every invariant of the prover must hold, and a broken gate must break the quotient and nothing else"""
    from tools import prove_flow

    res = prove_flow.run(K, 4, seed=22, verbose=False)
    assert all(res["checks"].values()), res["checks"]
    assert set(res["checks"]) >= {"quotient_is_a_polynomial", "permutation_product_closes", "lookup_product_closes", "commit_lagrange_equals_commit_coeff"}
    bad = prove_flow.run(K, 4, seed=22, corrupt="gate", verbose=False)["checks"]
    assert not bad["quotient_is_a_polynomial"] and bad["permutation_product_closes"] and bad["lookup_product_closes"] and bad["commit_lagrange_equals_commit_coeff"]


def wrapper_like_cs():
    """halo2-lib's BaseCircuitBuilder shape at the wrapper's parameters: 4 vertical-gate advice columns, one range-lookup column, a
    permutation over the advice, a fixed and the instance column (/root/reference/aggregator/src/wrapper.rs:792-797)"""
    A = 4
    gates = [[E.Fixed(i) * (E.Advice(i, 0) + E.Advice(i, 1) * E.Advice(i, 2) - E.Advice(i, 3))] for i in range(A)]
    lookups = [E.Lookup([E.Advice(A)], [E.Fixed(A)])]
    perm = [("advice", i) for i in range(A + 1)] + [("fixed", A + 1), ("instance", 0)]
    return E.ConstraintSystem(num_fixed=A + 2, num_advice=A + 1, num_instance=1, gates=gates, lookups=lookups, permutation_columns=perm,
                              blinding_factors=5, degree=4)


def test_evaluate_h_over_the_extended_coset_of_k22(lib):
    """2^24 rows of random column data resident in HBM; rows sampled everywhere plus the rows whose rotations wrap around the end of the
    coset, against the oracle's interpreter (the k = 14 test of tests/test_gpu_rows.py at the wrapper's size)"""
    import torch

    rows = 1 << EK
    cs = wrapper_like_cs()
    qc = E.quotient_columns(cs)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2204)
    d_cols = []
    for _ in range(qc.total):
        t = torch.randint(-(1 << 63), (1 << 63) - 1, (rows, 4), dtype=torch.int64, device="cuda", generator=gen)
        t[:, 3] = torch.randint(0, 0x30644E72E131A029, (rows,), dtype=torch.int64, device="cuda", generator=gen)      # canonical word patterns
        d_cols.append(t)
    d_out = torch.zeros(rows * 4, dtype=torch.int64, device="cuda")
    rng = random.Random(9)
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prog = E.evaluate_h_program(cs, K, EK, beta, gamma, theta, y)
    prog.run_device([t.data_ptr() for t in d_cols], EK, d_out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    scale = 1 << (EK - K)
    only = sorted(set(rng.sample(range(rows), 130) + [0, 1, 2, 3, scale - 1, scale, 3 * scale, rows - 1, rows - 2, rows - scale, rows - scale - 1,
                                                      rows - 3 * scale, rows - 6 * scale, rows // 2, rows // 2 - 1, (1 << 22) - 1, 1 << 22, (1 << 23) + 5]))
    need = set()
    for r in only:
        for rot in prog.rotations:
            need.add((r + rot * prog.rot_scale) % rows)
    need = sorted(need)
    idx = torch.tensor(need, dtype=torch.int64, device="cuda")
    sparse = [dict(zip(need, F.fr_decode(t[idx].cpu().numpy().view(np.uint64)))) for t in d_cols]
    exp = O.row_program_run(prog.insns, prog.constants, prog.rotations, prog.rot_scale, prog.result_reg, sparse, EK, omega=prog.omega, only_rows=only)
    got = d_out.view(rows, 4)[torch.tensor(only, dtype=torch.int64, device="cuda")].cpu().numpy().view(np.uint64)
    assert F.fr_decode(got) == exp


def test_grand_product_at_2p22_vs_the_c_oracle(lib, cref):
    """z[0] = 1, z[i+1] = z[i] num[i] / den[i] over 2^22 rows, device-resident (the entry point the prover flow uses), against
    ff::BatchInvert + the running product of the C oracle; a zero denominator late in the column zeroes everything after it"""
    import torch

    n = 1 << K
    num = cref.gen_scalars(2201, n, 0)
    den = cref.gen_scalars(2202, n, 0)
    den[n - 1000] = 0
    d_num = torch.from_numpy(num.view(np.int64)).cuda()
    d_den = torch.from_numpy(den.view(np.int64)).cuda()
    d_z = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_fr_grand_product_device(d_num.data_ptr(), d_den.data_ptr(), n, d_z.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    inv = den.copy()
    cref.batch_invert(inv)
    exp = cref.prefix_product(cref.field_op(1, 0, num, inv))
    got = d_z.cpu().numpy().view(np.uint64).reshape(n, 4)
    assert np.array_equal(got, exp)
    assert np.array_equal(d_num.cpu().numpy().view(np.uint64).reshape(n, 4), num)       # num is preserved (den is scratch: the header's contract)
    assert not got[n - 999:].any() and got[n - 1000].any()


def test_lookup_permute_at_2p22_vs_the_oracle(lib):
    """the wrapper's range lookup (lookup_bits = 21: /root/reference/aggregator/benches/wrapper_circuit.rs:66): 2^22 - 6 usable rows, inputs
    below 2^21, the table = the range (each value twice), shuffled; permuted input / table against the oracle's restatement"""
    import torch

    n, bits = 1 << K, 21
    usable = n - 6
    g = np.random.default_rng(2203)
    inputs = g.integers(0, 1 << bits, size=n, dtype=np.int64)
    table = (np.arange(n, dtype=np.int64) % (1 << bits))
    head = table[:usable].copy()
    g.shuffle(head)
    table[:usable] = head
    exp_in, exp_tab = O.permute_expression_pair_np(inputs, table, usable)

    def words(v):
        w = np.zeros((v.shape[0], 4), dtype=np.uint64)
        w[:, 0] = v.astype(np.uint64)
        return to_montgomery(lib, w)

    d_in = torch.from_numpy(words(inputs).view(np.int64)).cuda()
    d_tab = torch.from_numpy(words(table).view(np.int64)).cuda()
    d_pin = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    d_ptab = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    _lib.check(lib.zkhip_lookup_permute_device(d_in.data_ptr(), d_tab.data_ptr(), usable, d_pin.data_ptr(), d_ptab.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got_in = from_montgomery(lib, d_pin.cpu().numpy().view(np.uint64).reshape(n, 4)[:usable])
    got_tab = from_montgomery(lib, d_ptab.cpu().numpy().view(np.uint64).reshape(n, 4)[:usable])
    assert not got_in[:, 1:].any() and not got_tab[:, 1:].any()
    assert np.array_equal(got_in[:, 0].astype(np.int64), exp_in)
    assert np.array_equal(got_tab[:, 0].astype(np.int64), exp_tab)
