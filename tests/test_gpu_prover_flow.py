"""GPU (-m gpu): the §8(f) pieces composed the way create_proof composes them ([DEP] halo2-axiom plonk/prover.rs, reached from
/root/reference/aggregator/src/wrapper.rs:129), on a small satisfied circuit of the halo2-lib shape: vertical gate
q (a + b c - d), a range lookup, copy constraints over advice and fixed columns.

Parity at this boundary is unpinned by the reference (no fixtures), and evaluate_h is restated from the published algorithm; this
test therefore pins the restatement to the *mathematics* instead: for a satisfying witness the quotient numerator built by the
row program must be divisible by X^n - 1, i.e. after divide_by_vanishing_poly and the inverse extended transform every
coefficient of degree >= 3n vanishes -- and it must stop vanishing as soon as one gate output, one copy, or one grand product is
broken.  Along the way: both grand products close (z(omega^u) = 1) and the permuted lookup pair satisfies its defining property."""
import random

import numpy as np
import pytest

import zksnap_circuits_halo2_amd as Z
from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F

pytestmark = pytest.mark.gpu
R = O.R_MOD
K, BLIND = 6, 5
N = 1 << K
U = N - (BLIND + 1)                       # usable rows
PERM_COLUMNS = [("advice", 0), ("advice", 1), ("fixed", 1)]


def enc(v):
    return F.fr_encode(v)


def toy_circuit(rng, break_gate=False, break_copy=False):
    """fixed [q0, fconst, table], advice [a0 (gate column), a1 (lookup input)]; returns the Lagrange columns and sigma"""
    q0 = [1 if (i % 4 == 0 and i + 3 < U) else 0 for i in range(N)]
    table = [i % 16 for i in range(N)]
    fconst = [rng.randrange(R) for _ in range(N)]
    a0 = [rng.randrange(R) for _ in range(N)]
    a1 = [rng.randrange(16) if i < U else rng.randrange(R) for i in range(N)]
    # copy constraints between free cells (gate inputs, lookup inputs, fixed constants): cycles of (column, row)
    cycles = [[(0, 1), (2, 2)], [(1, 10), (1, 20)], [(0, 13), (1, 30)], [(0, 17), (0, 21), (2, 5)]]
    cols = [a0, a1, fconst]
    for cyc in cycles:
        lookup_rows = [r for c, r in cyc if c == 1]
        v = a1[lookup_rows[0]] if lookup_rows else cols[cyc[0][0]][cyc[0][1]]     # a cycle through a lookup cell carries a table value
        for c, r in cyc:
            cols[c][r] = v
    if break_copy:
        a0[21] = (a0[21] + 1) % R                           # before the gate outputs: only the copy is broken
    for i in range(N):
        if q0[i]:
            a0[i + 3] = (a0[i] + a0[i + 1] * a0[i + 2]) % R
    if break_gate:
        a0[7] = (a0[7] + 1) % R
    # sigma: identity, then each cycle maps a cell to the next one
    sigma = {(c, r): (c, r) for c in range(3) for r in range(N)}
    for cyc in cycles:
        for idx, cell in enumerate(cyc):
            sigma[cell] = cyc[(idx + 1) % len(cyc)]
    omega = O.omega_for(K)
    sig_cols = [[pow(O.FR_DELTA, sigma[(c, r)][0], R) * pow(omega, sigma[(c, r)][1], R) % R for r in range(N)] for c in range(3)]
    return dict(fixed=[q0, fconst, table], advice=[a0, a1], sigma=sig_cols)


def quotient_top_coefficients(circ, rng, tamper_z=False):
    cs = E.ConstraintSystem(num_fixed=3, num_advice=2, num_instance=0,
                            gates=[[E.Fixed(0) * (E.Advice(0, 0) + E.Advice(0, 1) * E.Advice(0, 2) - E.Advice(0, 3))]],
                            lookups=[E.Lookup([E.Advice(1)], [E.Fixed(2)])], permutation_columns=PERM_COLUMNS,
                            blinding_factors=BLIND, degree=4)
    dom = Z.EvaluationDomain(4, K)
    ek = dom.extended_k
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    lagr = {"fixed": circ["fixed"], "advice": circ["advice"]}
    values = [lagr[kind][idx] for kind, idx in PERM_COLUMNS]
    blind = lambda col: col[:U + 1] + [rng.randrange(R) for _ in range(N - U - 1)]

    # permutation argument: one z per chunk of chunk_len columns, chained through z_k[0] = z_{k-1}[u]
    z_sets, last_z = [], 1
    for s in range(cs.num_permutation_sets):
        lo, hi = s * cs.chunk_len, min((s + 1) * cs.chunk_len, len(values))
        w = hi - lo
        num = E.permutation_numerator_program(w, lo, beta, gamma, K).run([enc(c) for c in values[lo:hi]], K)
        den = E.permutation_denominator_program(w, beta, gamma).run([enc(c) for c in values[lo:hi]] + [enc(c) for c in circ["sigma"][lo:hi]], K)
        z = np.zeros((N, 4), dtype=np.uint64)
        _lib.check(_lib.load().zkhip_fr_grand_product(num.ctypes.data, den.ctypes.data, N, z.ctypes.data))
        z = [v * last_z % R for v in F.fr_decode(z)]
        last_z = z[U]
        z_sets.append(blind(z))
    perm_closes = last_z == 1

    # lookup argument
    A, S = circ["advice"][1], circ["fixed"][2]
    pa, ps = E.permute_expression_pair(enc(A), enc(S), U)
    pa = F.fr_decode(pa) + [rng.randrange(R) for _ in range(N - U)]
    ps = F.fr_decode(ps) + [rng.randrange(R) for _ in range(N - U)]
    for i in range(U):
        assert ps[i] == pa[i] or (i > 0 and pa[i] == pa[i - 1])
    pn, pd = E.lookup_product_programs(1, 1, beta, gamma, theta)
    num, den = pn.run([enc(A), enc(S)], K), pd.run([enc(pa), enc(ps)], K)
    zl = np.zeros((N, 4), dtype=np.uint64)
    _lib.check(_lib.load().zkhip_fr_grand_product(num.ctypes.data, den.ctypes.data, N, zl.ctypes.data))
    zl = F.fr_decode(zl)
    lookup_closes = zl[U] == 1
    zl = blind(zl)
    if tamper_z:
        z_sets[0][9] = (z_sets[0][9] + 1) % R

    l0 = [1 if i == 0 else 0 for i in range(N)]
    l_last = [1 if i == U else 0 for i in range(N)]
    l_active = [1 if i < U else 0 for i in range(N)]
    qc = E.quotient_columns(cs)
    lagrange_cols = circ["fixed"] + circ["advice"] + [l0, l_last, l_active] + circ["sigma"] + z_sets + [zl, pa, ps]
    assert len(lagrange_cols) == qc.total
    ext_cols = [dom.coeff_to_extended(dom.lagrange_to_coeff(enc(c))) for c in lagrange_cols]

    prog = E.evaluate_h_program(cs, K, ek, beta, gamma, theta, y)
    numerator = prog.run(ext_cols, ek)
    h_ext = dom.divide_by_vanishing_poly(numerator)
    coeffs = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
    buf = h_ext.copy()
    _lib.check(_lib.load().zkhip_extended_to_coeff(buf.ctypes.data, ek, dom.extended_omega_inv.ctypes.data, dom.extended_ifft_divisor.ctypes.data,
                                                   dom.g_coset.ctypes.data, coeffs.ctypes.data, dom.extended_len()))
    c = F.fr_decode(coeffs)
    return c[:3 * N], c[3 * N:], perm_closes, lookup_closes


def test_satisfied_circuit_gives_a_polynomial_quotient(lib):
    rng = random.Random(2024)
    low, top, perm_closes, lookup_closes = quotient_top_coefficients(toy_circuit(rng), rng)
    assert perm_closes and lookup_closes
    assert any(low) and not any(top)                      # h has degree < 3n: the numerator is a multiple of X^n - 1


@pytest.mark.parametrize("what", ["gate", "copy", "z"])
def test_broken_witness_is_not_divisible(lib, what):
    rng = random.Random(7)
    circ = toy_circuit(rng, break_gate=what == "gate", break_copy=what == "copy")
    low, top, perm_closes, lookup_closes = quotient_top_coefficients(circ, rng, tamper_z=what == "z")
    assert any(top)
    assert perm_closes == (what != "copy")                # a broken copy also leaves the permutation product open
    assert lookup_closes


def test_lookup_input_outside_the_table_is_reported(lib):
    rng = random.Random(9)
    circ = toy_circuit(rng)
    circ["advice"][1][3] = 99                              # not a table value
    with pytest.raises(_lib.ZkhipError):
        quotient_top_coefficients(circ, rng)


@pytest.mark.parametrize("k,gate_cols", [(7, 1), (10, 3)])
def test_device_resident_prover_flow(lib, k, gate_cols):
    """tools/prove_flow.py: the same composition on device-resident columns with the batched entry points and a structured SRS
    (setup, advice commitments in the Lagrange basis, permutation / lookup arguments, batched iNTT / coset NTT, fused quotient, h
    commitments, batched evaluations); the prover's invariants must hold, and must break with the witness"""
    from tools import prove_flow

    res = prove_flow.run(k, gate_cols, seed=5 + k, verbose=False)
    assert all(res["checks"].values()), res["checks"]
    bad = prove_flow.run(k, gate_cols, seed=5 + k, corrupt="gate", verbose=False)["checks"]
    assert not bad["quotient_is_a_polynomial"] and bad["permutation_product_closes"] and bad["commit_lagrange_equals_commit_coeff"]
    bad = prove_flow.run(k, gate_cols, seed=5 + k, corrupt="copy", verbose=False)["checks"]
    assert not bad["quotient_is_a_polynomial"] and not bad["permutation_product_closes"]


@pytest.mark.parametrize("k,gate_cols,lookups", [(7, 3, 2), (13, 256, 8), (15, 64, 8)])
def test_device_resident_prover_flow_wide(lib, k, gate_cols, lookups):
    """the small circuits' REAL shape: the standalone voter / state-transition benches size their columns with
    `calculate_params(Some(20))` (/root/reference/voter/benches/voter_circuit.rs:49-51,
    /root/reference/aggregator/benches/state_transition_circuit.rs:48-50; browser config 412 advice + 11 lookup columns,
    /root/reference/voter/frontend/app/worker.js:95-102) -- hundreds of gate columns and several lookup columns, where the tests above use
    <= 4 and one.  Same flow, same invariants, all the columns of a phase committed through the batched registered-bases entry point; the
    quotient program then names hundreds of columns and runs in thousands of instructions."""
    from tools import prove_flow

    res = prove_flow.run(k, gate_cols, seed=70 + k, verbose=False, lookups=lookups)
    assert all(res["checks"].values()), res["checks"]
    assert res["columns"] == (gate_cols + 2) + (gate_cols + lookups) + 3 + (gate_cols + lookups + 1) + -(-(gate_cols + lookups + 1) // 2) + 3 * lookups
    assert res["program_registers"] <= _lib.VM_REGS and res["program_insns"] > 4 * gate_cols
    bad = prove_flow.run(k, gate_cols, seed=70 + k, corrupt="gate", verbose=False, lookups=lookups)["checks"]
    assert not bad["quotient_is_a_polynomial"] and bad["permutation_product_closes"] and bad["lookup_product_closes"]
    if k <= 13:
        bad = prove_flow.run(k, gate_cols, seed=70 + k, corrupt="copy", verbose=False, lookups=lookups)["checks"]
        assert not bad["quotient_is_a_polynomial"] and not bad["permutation_product_closes"]


# ---------------------------------------------------------------- keygen_vk / keygen_pk, ProvingKey files
def test_keygen_against_the_oracle(lib, cref):
    """keygen.py on the toy circuit: sigma columns = the cycles' delta^col * omega^row; the verifying key's commitments = the oracle's
    MSM of the Lagrange columns; fixed_polys / fixed_cosets / permutation cosets = the oracle's `lagrange_to_coeff` / `coeff_to_extended`;
    l_active_row = 1 - (l_last + l_blind) on the extended coset, the formula `keygen_pk` [DEP plonk/keygen.rs] evaluates"""
    from zksnap_circuits_halo2_amd import keygen as KG

    rng = random.Random(31)
    circ = toy_circuit(rng)
    cs = E.ConstraintSystem(num_fixed=3, num_advice=2, gates=[[E.Fixed(0) * (E.Advice(0, 0) + E.Advice(0, 1) * E.Advice(0, 2) - E.Advice(0, 3))]],
                            lookups=[E.Lookup([E.Advice(1)], [E.Fixed(2)])], permutation_columns=PERM_COLUMNS, blinding_factors=BLIND, degree=4)
    asm = KG.Assembly(N, 3)
    for cyc in [[(0, 1), (2, 2)], [(1, 10), (1, 20)], [(0, 13), (1, 30)], [(0, 17), (0, 21), (2, 5)]]:
        for (c1, r1), (c2, r2) in zip(cyc, cyc[1:]):
            asm.copy(c1, r1, c2, r2)
    fixed = [enc(c) for c in circ["fixed"]]
    with Z.ParamsKZG.setup(K, 0xC0FFEE) as params:
        vk = KG.keygen_vk(params, cs, fixed, asm)
        pk = KG.keygen_pk(params, vk, cs, fixed, asm)
        gl = params.g_lagrange.copy()
    for c in range(3):
        assert F.fr_decode(pk.permutations[c]) == circ["sigma"][c], c
    for cols, commits in ((fixed, vk.fixed_commitments), (pk.permutations, vk.permutation_commitments)):
        for col, cm in zip(cols, commits):
            assert np.array_equal(cm, cref.jac_to_affine(cref.best_multiexp(np.ascontiguousarray(col), gl, 2)))
    dom = Z.EvaluationDomain(4, K)

    def to_coeff(col):
        a = np.ascontiguousarray(col).copy()
        cref.best_fft(a, dom.omega_inv, K, 1)
        cref.scale(a, dom.ifft_divisor)
        return a

    def to_ext(coeff):
        ext = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
        ext[:N] = coeff
        cref.distribute_powers_zeta(ext[:N], dom.g_coset, dom.g_coset_inv)
        cref.best_fft(ext, dom.extended_omega, dom.extended_k, 1)
        return ext

    for lag, poly, coset in list(zip(fixed, pk.fixed_polys, pk.fixed_cosets)) + list(zip(pk.permutations, pk.permutation_polys, pk.permutation_cosets)):
        assert np.array_equal(poly, to_coeff(lag))
        assert np.array_equal(coset, to_ext(poly))
    ind = lambda rows: enc([1 if i in rows else 0 for i in range(N)])
    assert np.array_equal(pk.l0, to_ext(to_coeff(ind({0}))))
    l_last, l_blind = to_ext(to_coeff(ind({U}))), to_ext(to_coeff(ind(set(range(U + 1, N)))))
    assert np.array_equal(pk.l_last, l_last)
    want = [(1 - a - b) % R for a, b in zip(F.fr_decode(l_last), F.fr_decode(l_blind))]
    assert F.fr_decode(pk.l_active_row) == want


@pytest.mark.parametrize("k,gate_cols", [(7, 1), (10, 3)])
def test_prover_flow_from_a_proving_key_file(lib, tmp_path, k, gate_cols):
    """keygen -> `ProvingKey::write(RawBytesUnchecked)` -> `ProvingKey::read` -> the prover runs on the key that was READ (what the
    reference's wrapper does through build/*_pk.bin: /root/reference/aggregator/src/wrapper.rs:967-989, :1007-1034); the prover's
    invariants hold, and still break with the witness"""
    from tools import prove_flow

    path = str(tmp_path / "toy_pk.bin")
    res = prove_flow.run(k, gate_cols, seed=40 + k, verbose=False, pk_file=path)
    assert all(res["checks"].values()), res["checks"]
    n, en, nf, npc = 1 << k, 1 << (k + 2), gate_cols + 2, gate_cols + 2
    want = (8 + nf * 64 + npc * 64) + 3 * (4 + en * 32) + 2 * (4 + nf * (4 + n * 32)) + (4 + nf * (4 + en * 32)) + 2 * (4 + npc * (4 + n * 32)) + (4 + npc * (4 + en * 32))
    assert res["pk_file_bytes"] == want
    bad = prove_flow.run(k, gate_cols, seed=40 + k, corrupt="copy", verbose=False, pk_file=path)["checks"]
    assert not bad["quotient_is_a_polynomial"] and not bad["permutation_product_closes"]


def test_params_file_reader_checks_every_point(lib, cref, tmp_path):
    """`ParamsKZG::read` (SerdeFormat::RawBytes) verifies every point: one off-curve point anywhere in either table is refused by the
    default reader (a GPU pass over the whole file), found at its index, and accepted only by the explicit unchecked mode"""
    import io

    from zksnap_circuits_halo2_amd import srs

    k = 9
    with Z.ParamsKZG.setup(k, 777) as params:
        buf = io.BytesIO()
        params.write(buf)
    raw = bytearray(buf.getvalue())
    srs.read_params(io.BytesIO(bytes(raw)))                                    # intact file: accepted
    idx = 301                                                                  # not one of 64 evenly spaced samples
    off = 4 + (1 << k) * 64 + idx * 64 + 32                                    # y of g_lagrange[idx]
    raw[off] ^= 1
    with pytest.raises(ValueError, match=r"g_lagrange: point 301"):
        srs.read_params(io.BytesIO(bytes(raw)))
    srs.read_params(io.BytesIO(bytes(raw)), check_points=0)                    # RawBytesUnchecked
    pts = np.frombuffer(bytes(raw[4:4 + (1 << k) * 64]), dtype="<u8").reshape(-1, 8).astype(np.uint64)
    assert srs.g1_first_invalid(pts) is None
    pts[7, 0:4] = np.array([(O.Q_MOD >> (64 * j)) & ((1 << 64) - 1) for j in range(4)], dtype=np.uint64)     # x = q: not canonical
    assert srs.g1_first_invalid(pts) == 7


def test_cpp_keygen_mirror_matches_python(lib, cref, tmp_path):
    """The compiled-host mirror (include/zkhip.hpp: Assembly, keygen_vk / keygen_pk, ProvingKey::write / read, best_multiexp::<G2Affine>,
    set_msm_shards) against the Python mirror: the two proving-key files of the same circuit are byte-identical, each side reads the other's
    file, the C++ G2 MSM equals the oracle's, and commits agree across shard counts."""
    import io
    import os
    import struct
    import subprocess

    from tests.test_gpu_g2 import dec as g2_dec, enc as g2_enc, walk as g2_walk
    from zksnap_circuits_halo2_amd import keygen as KG

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "tests", "cpp", "keygen_driver")
    assert os.path.exists(drv), "build it with __graft_entry__.build()"
    k, trapdoor = K, 0xABCDEF12345
    rng = random.Random(77)
    circ = toy_circuit(rng)
    fixed = [enc(c) for c in circ["fixed"]]
    copies = [(0, 1, 2, 2), (1, 10, 1, 20), (0, 13, 1, 30), (0, 17, 0, 21), (0, 21, 2, 5), (2, 40, 0, 41), (0, 41, 2, 40)]
    m = 200
    pts = g2_enc(g2_walk(31337, 4242, m))
    sc = cref.gen_scalars(7701, m, 0)
    fin = tmp_path / "in.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<IIIQ", k, 3, 3, trapdoor))
        for col in fixed:
            f.write(np.ascontiguousarray(col).tobytes())
        f.write(struct.pack("<I", len(copies)))
        for c in copies:
            f.write(struct.pack("<IIII", *c))
        f.write(struct.pack("<I", m))
        f.write(pts.tobytes())
        f.write(sc.tobytes())
    # Python side first: its key file is handed to the C++ reader
    cs = E.ConstraintSystem(num_fixed=3, num_advice=2, permutation_columns=PERM_COLUMNS, blinding_factors=BLIND, degree=4)
    asm = KG.Assembly(N, 3)
    for c in copies:
        asm.copy(*c)
    with Z.ParamsKZG.setup(k, trapdoor) as params:
        vk = KG.keygen_vk(params, cs, fixed, asm)
        pk = KG.keygen_pk(params, vk, cs, fixed, asm)
    buf = io.BytesIO()
    pk.write(buf, KG.RAW_BYTES_UNCHECKED)
    py_file = tmp_path / "py_pk.bin"
    py_file.write_bytes(buf.getvalue())
    cpp_file, report = tmp_path / "cpp_pk.bin", tmp_path / "report.bin"
    # SerdeFormat::Processed on both sides: the key and the parameter file (compressed points, canonical scalars)
    pbuf = io.BytesIO()
    pk.write(pbuf, KG.PROCESSED)
    with Z.ParamsKZG.setup(k, trapdoor) as params:
        raw_params, proc_params = io.BytesIO(), io.BytesIO()
        params.write(raw_params)
        params.write_custom(proc_params, KG.PROCESSED)
    py_params = tmp_path / "py_params.bin"
    py_params.write_bytes(raw_params.getvalue())
    cpp_pk_proc, cpp_params_proc = tmp_path / "cpp_pk_processed.bin", tmp_path / "cpp_params_processed.bin"
    subprocess.check_call([drv, str(fin), str(cpp_file), str(report), str(py_file), str(cpp_pk_proc), str(py_params), str(cpp_params_proc)], timeout=300)
    assert cpp_file.read_bytes() == buf.getvalue(), "the C++ and Python mirrors write different proving-key files"
    assert cpp_pk_proc.read_bytes() == pbuf.getvalue(), "the two mirrors write different SerdeFormat::Processed key files"
    assert cpp_params_proc.read_bytes() == proc_params.getvalue(), "the two mirrors write different SerdeFormat::Processed parameter files"
    assert len(pbuf.getvalue()) < len(buf.getvalue()) and len(proc_params.getvalue()) == 4 + 2 * N * 32 + 128
    back = KG.ProvingKey.read(io.BytesIO(cpp_file.read_bytes()), KG.RAW_BYTES, cs)      # checked reader on the C++ file
    assert np.array_equal(back.vk.fixed_commitments, vk.fixed_commitments) and np.array_equal(back.permutation_cosets[2], pk.permutation_cosets[2])
    rep = report.read_bytes()
    flags, = struct.unpack("<Q", rep[:8])
    assert flags == 0b111111, bin(flags)
    g2 = np.frombuffer(rep[8:8 + 192], dtype=np.uint64)
    assert g2_dec(g2) == O.g2_scalar_mul(cref.expected_scalar(sc, 31337, 4242), O.G2_GEN)


def test_device_resident_keygen_equals_host_keygen(lib, cref):
    """`keygen_device` (sigma columns by gather in HBM, commitments against the registered g_lagrange through
    `zkhip_msm_g1_registered_device`, transforms in place) produces the key of `keygen_vk` + `keygen_pk` element for element; the
    uploaded form of a key that was read from a file is the same memory"""
    import ctypes as C
    import io

    from zksnap_circuits_halo2_amd import keygen as KG

    rng = random.Random(91)
    circ = toy_circuit(rng)
    fixed = [enc(c) for c in circ["fixed"]]
    cs = E.ConstraintSystem(num_fixed=3, num_advice=2, permutation_columns=PERM_COLUMNS, blinding_factors=BLIND, degree=4)
    asm = KG.Assembly(N, 3)
    for c in [(0, 1, 2, 2), (1, 10, 1, 20), (0, 13, 1, 30), (0, 17, 0, 21), (0, 21, 2, 5), (2, 40, 0, 41)]:
        asm.copy(*c)
    with Z.ParamsKZG.setup(K, 0xD15C0) as params:
        vk = KG.keygen_vk(params, cs, fixed, asm)
        pk = KG.keygen_pk(params, vk, cs, fixed, asm)
        with KG.keygen_device(params, cs, fixed, asm) as dpk:
            host = dpk.to_host()
            # the accessors address what to_host() downloads
            one = np.zeros((N, 4), dtype=np.uint64)
            _lib.check(lib.zkhip_download(one.ctypes.data, C.c_void_p(dpk.permutation_poly(2)), N * 32))
            assert np.array_equal(one, pk.permutation_polys[2])
        # the registered-bases device entry point on its own: device scalars, host pointer of the registered table, any sub-range
        sc = cref.gen_scalars(4411, N, 0)
        d = C.c_void_p()
        _lib.check(lib.zkhip_alloc(N * 32 + 96, C.byref(d)))
        try:
            _lib.check(lib.zkhip_upload(d, sc.ctypes.data, N * 32))
            d_out = C.c_void_p(d.value + N * 32)
            for lo, m in ((0, N), (5, 37), (N - 1, 1), (3, 0)):
                _lib.check(lib.zkhip_msm_g1_registered_device(params.g_lagrange[lo:].ctypes.data, C.c_void_p(d.value + lo * 32), m, d_out, None))
                got = np.zeros(12, dtype=np.uint64)
                _lib.check(lib.zkhip_download(got.ctypes.data, d_out, 96))
                exp = cref.best_multiexp(np.ascontiguousarray(sc[lo:lo + m]), np.ascontiguousarray(params.g_lagrange[lo:lo + m]), 2)
                assert np.array_equal(cref.jac_to_affine(got), cref.jac_to_affine(exp))
            other = np.zeros((N, 8), dtype=np.uint64)
            assert lib.zkhip_msm_g1_registered_device(other.ctypes.data, d, N, d_out, None) == -1      # not a registered array
        finally:
            lib.zkhip_free(d)
    a, b = io.BytesIO(), io.BytesIO()
    pk.write(a)
    host.write(b)
    assert a.getvalue() == b.getvalue()
    with KG.DeviceProvingKey.from_host(pk, cs) as up:
        c = io.BytesIO()
        up.to_host().write(c)
        assert c.getvalue() == a.getvalue()


def test_gather_mul_vs_big_integers(lib):
    import ctypes as C

    rng = random.Random(17)
    na, nb, n = 37, 5, 1000
    a = [rng.randrange(R) for _ in range(na)]
    b = [rng.randrange(R) for _ in range(nb)]
    ia = np.array([rng.randrange(na) for _ in range(n)], dtype=np.uint32)
    ib = np.array([rng.randrange(nb) for _ in range(n)], dtype=np.uint32)
    ia[0], ib[0] = na + 3, nb                                                       # out of range: reduced modulo the table length
    bufs = [C.c_void_p() for _ in range(5)]
    data = [enc(a), ia, enc(b), ib, np.zeros((n, 4), dtype=np.uint64)]
    try:
        for p, arr in zip(bufs, data):
            _lib.check(lib.zkhip_alloc(arr.nbytes, C.byref(p)))
            _lib.check(lib.zkhip_upload(p, arr.ctypes.data, arr.nbytes))
        _lib.check(lib.zkhip_fr_gather_mul_device(bufs[0], na, bufs[1], bufs[2], nb, bufs[3], n, bufs[4], None))
        out = np.zeros((n, 4), dtype=np.uint64)
        _lib.check(lib.zkhip_download(out.ctypes.data, bufs[4], out.nbytes))
        assert F.fr_decode(out) == [a[int(i) % na] * b[int(j) % nb] % R for i, j in zip(ia, ib)]
        assert lib.zkhip_fr_gather_mul_device(bufs[0], 0, bufs[1], bufs[2], nb, bufs[3], n, bufs[4], None) == -1
    finally:
        for p in bufs:
            lib.zkhip_free(p)
