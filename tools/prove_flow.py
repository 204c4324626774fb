#!/usr/bin/env python3
"""A transcript-less `create_proof` for a *satisfied* circuit of the halo2-lib shape, device-resident from witness columns to
quotient commitments -- the steps of [DEP] halo2-axiom plonk/prover.rs in the order the reference's prover runs them
(/root/reference/aggregator/src/wrapper.rs:129), composed from this repo's entry points only:

  SRS (ParamsKZG.setup, known trapdoor) -> advice commitments (Lagrange basis) -> permutation and lookup arguments (row programs,
  permute_expression_pair, grand products) -> Lagrange -> coefficients (batched iNTT) -> commitments -> extended coset (batched NTT)
  -> fused quotient program -> / (X^n - 1) -> inverse extended transform -> h commitments -> evaluations at x.

Circuit: `gate_cols` advice columns with the vertical gate q (a + b c - d) on rows 0, 4, 8, ..., `lookups` range-lookup columns against a
2^bits-entry table, copy constraints across advice / fixed columns, 5 blinding rows.  The small circuits of the reference size their columns
with `calculate_params(Some(20))` (/root/reference/voter/benches/voter_circuit.rs:49-51,
/root/reference/aggregator/benches/state_transition_circuit.rs:48-50; the browser config is 412 advice + 11 lookup columns at k = 15:
/root/reference/voter/frontend/app/worker.js:95-102): `run(13, 256, lookups=8)` and `run(15, 64, lookups=8)` are those shapes, with all
the columns of a phase committed through ONE batched call (`zkhip_msm_g1_registered_batch_device`) as a Rust host would have to.  Checks (the prover's own invariants): both
grand products close, the quotient is a polynomial (coefficients of degree >= 3n vanish), commit_lagrange(column) = commit(coefficients).
There is no transcript: challenges are seeded.  Usage: prove_flow.py [k] [gate_cols] [lookups]   (default 16 4 1)."""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch

import zksnap_circuits_halo2_amd as Z
from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F

R = F.R_MOD
BLIND = 5


_SIDE_STREAM = None


def run(k=16, gate_cols=4, seed=1, lookup_bits=8, corrupt=None, verbose=True, pk_file=None, lookups=1, batched=None):
    """pk_file: path -- the proving key is written there (`ProvingKey::write`, RawBytesUnchecked), read back, and the READ key is what the
    prover uses (the reference's wrapper does the same through build/*_pk.bin: /root/reference/aggregator/src/wrapper.rs:967-989, :1007-1034)"""
    from zksnap_circuits_halo2_amd import keygen as KG

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, u = 1 << k, (1 << k) - (BLIND + 1)
    lookup_bits = min(lookup_bits, k - 1)            # every table value must occur among the usable rows
    rng = random.Random(seed)
    torch.manual_seed(seed)
    beta, gamma, theta, y, x, s = (rng.randrange(1, R) for _ in range(6))
    dom = Z.EvaluationDomain(4, k)
    ek, en = dom.extended_k, dom.extended_len()
    G, NL = gate_cols, lookups
    if batched is None:
        batched = k <= 17                # small MSMs are latency-bound one at a time: all columns of a phase in one launch set
    t = {}
    clock = [time.perf_counter()]

    def lap(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        t[name] = t.get(name, 0.0) + (now - clock[0]) * 1e3
        clock[0] = now

    def words(vals):                     # python ints -> device tensor of Montgomery words
        return torch.from_numpy(F.fr_encode(vals).view(np.int64)).to(dev)

    ONE = words([1])[0]

    dpk = None
    def rand_fr(m):                      # uniformly random canonical word patterns = random field elements
        a = torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, device=dev)
        a[:, 3] = torch.randint(0, 1 << 61, (m,), dtype=torch.int64, device=dev)
        return a

    def run_prog(prog, cols, log_rows, out=None):
        if out is None:
            out = torch.empty(((1 << log_rows), 4), dtype=torch.int64, device=dev)
        prog.run_device([c.data_ptr() for c in cols], log_rows, out.data_ptr())
        return out

    to_mont = E.RowProgram()             # raw integer words are the Montgomery form of a / R: multiply by R
    to_mont.emit(E.OP_MUL, 0, to_mont.column(0), to_mont.constant(pow(2, 256, R)))

    def small_ints(v):                   # int64 tensor of small non-negative integers -> Montgomery words
        a = torch.zeros((v.shape[0], 4), dtype=torch.int64, device=dev)
        a[:, 0] = v
        return run_prog(to_mont, [a], k)

    rows = torch.arange(n, dtype=torch.int64, device=dev)
    clock[0] = time.perf_counter()
    # ---- SRS -------------------------------------------------------------------------------------------------------------
    params = Z.ParamsKZG.setup(k, s)
    lap("setup_srs")
    h_g, h_gl = False, True                                                       # which of the params' registered tables a commit uses

    def commit(lagrange, col):                                                     # params.commit / commit_lagrange of a device-resident column
        out = torch.zeros(12, dtype=torch.int64, device=dev)
        params.commit_device(col.data_ptr(), n, out.data_ptr(), lagrange=lagrange)
        return out

    # the current stream and one high-priority stream: streams of different priority never share a hardware queue (two streams of one
    # priority may, and then run one after the other)
    global _SIDE_STREAM
    if _SIDE_STREAM is None:
        _SIDE_STREAM = torch.cuda.Stream(priority=-1)       # kept between proofs, like the scratch set the library keys by it
    side = [torch.cuda.current_stream(), _SIDE_STREAM]

    def commit_all(lagrange, cols):
        """independent commits alternate between two streams: one MSM's latency-bound reduction tail runs under the next one's accumulation
        (tools/two_stream_msm.py: 5.31 -> 4.96 ms per 2^22 MSM); each stream has its own scratch set inside the library.
        batched (small k): the columns are gathered into one array and committed by one call"""
        if batched and len(cols) > 1:
            stack = torch.stack(list(cols)).contiguous()
            outs_b = torch.zeros((len(cols), 12), dtype=torch.int64, device=dev)
            params.commit_many_device(stack.data_ptr(), n, len(cols), n, outs_b.data_ptr(), lagrange=lagrange)
            return [outs_b[i] for i in range(len(cols))]
        outs = [torch.zeros(12, dtype=torch.int64, device=dev) for _ in cols]
        cur = torch.cuda.current_stream()
        side[1].wait_stream(cur)
        for i, col in enumerate(cols):
            params.commit_device(col.data_ptr(), n, outs[i].data_ptr(), lagrange=lagrange, stream=side[i % 2].cuda_stream)
        cur.wait_stream(side[1])
        return outs

    def affine(jac):
        return F.g1_decode_jacobian(jac.cpu().numpy().view(np.uint64))

    try:
        # ---- fixed and advice columns (Lagrange basis) ---------------------------------------------------------------------
        gate_rows = (rows % 4 == 0) & (rows + 3 < u)
        sel = torch.where(gate_rows[:, None], ONE[None, :], torch.zeros_like(ONE)[None, :]).contiguous()
        sel3 = torch.roll(sel, 3, 0).contiguous()                                  # 1 on the gates' output rows
        fixed = [sel.clone() for _ in range(G)] + [rand_fr(n), small_ints(rows % (1 << lookup_bits))]        # q_0.., fconst, table
        fconst, table = fixed[G], fixed[G + 1]
        advice = [rand_fr(n) for _ in range(G)]
        lks = []
        for _ in range(NL):
            lk = small_ints(torch.randint(0, 1 << lookup_bits, (n,), dtype=torch.int64, device=dev))
            lk[u:] = rand_fr(n - u)
            lks.append(lk)
            advice.append(lk)
        FC = G + NL                                                                # index of the constants column among the permutation columns
        perm_cols = [("advice", i) for i in range(G + NL)] + [("fixed", G)]        # every advice column and the constants column
        pcol = lambda c: advice[c] if c < FC else fconst
        cycles = [[(0, 1), (FC, 2)], [(G, 10), (G, 20)], [(0, 13), (G, 30)], [(0, 17), (0, 21), (FC, 5)]]
        cycles += [[(G + j, 40 + j), (G, 60 + j)] for j in range(1, NL)]           # lookup column j <-> lookup column 0
        if G > 1:
            cycles += [[(G - 1, 5), (FC, 7)], [(G // 2, 9), (G - 1, 25)]]          # the far gate columns take part in the permutation too
        for cyc in cycles:                # equal values along every cycle (a cycle through a lookup column carries a table value)
            src = next(((c, r) for c, r in cyc if G <= c < FC), cyc[0])
            v = pcol(src[0])[src[1]].clone()
            for c, r in cyc:
                pcol(c)[r] = v
        gate = E.RowProgram()             # out = a + sel3 * ((a[-3] + a[-2] a[-1]) - a): the gate outputs, everything else unchanged
        gate.emit(E.OP_MUL, 0, gate.column(0, -2), gate.column(0, -1))
        gate.emit(E.OP_ADD, 0, E.RowProgram.reg(0), gate.column(0, -3))
        gate.emit(E.OP_SUB, 0, E.RowProgram.reg(0), gate.column(0, 0))
        gate.emit(E.OP_MAD, 0, E.RowProgram.reg(0), gate.column(1, 0), gate.column(0, 0))
        for i in range(G):
            advice[i] = run_prog(gate, [advice[i], sel3], k)
        if corrupt == "gate":
            advice[0][7] = advice[0][8].clone()
        if corrupt == "copy":
            advice[0][21] = advice[0][22].clone()
        lap("witness_columns")
        adv_commit = commit_all(h_gl, advice)                                       # advice is committed in the Lagrange basis
        lap("commit_advice")

        # ---- keygen: the circuit's fixed columns and copy constraints -> verifying key, proving key -----------------------------------
        omega = F.omega_for(k)
        cs = E.ConstraintSystem(
            num_fixed=G + 2, num_advice=G + NL, num_instance=0,
            gates=[[E.Fixed(i) * (E.Advice(i, 0) + E.Advice(i, 1) * E.Advice(i, 2) - E.Advice(i, 3))] for i in range(G)],
            lookups=[E.Lookup([E.Advice(G + j)], [E.Fixed(G + 1)]) for j in range(NL)], permutation_columns=perm_cols, blinding_factors=BLIND, degree=4)
        assembly = KG.Assembly(n, len(perm_cols))
        for cyc in cycles:
            for (c1, r1), (c2, r2) in zip(cyc, cyc[1:]):
                assembly.copy(c1, r1, c2, r2)
        host = lambda tns: tns.cpu().numpy().view(np.uint64).reshape(-1, 4)
        lap("witness_columns")
        pk_bytes = None
        if pk_file is None:
            # the key is produced in HBM and stays there (keygen.keygen_device): sigma columns by gather, commitments against the
            # registered g_lagrange, transforms in place
            dpk = KG.keygen_device(params, cs, [host(f) for f in fixed], assembly)
            vk = dpk.vk
            lap("keygen_device")
        else:
            vk = KG.keygen_vk(params, cs, [host(f) for f in fixed], assembly)
            lap("keygen_vk")
            pk = KG.keygen_pk(params, vk, cs, [host(f) for f in fixed], assembly)
            lap("keygen_pk")
        if pk_file is not None:
            with open(pk_file, "wb") as fh:
                pk.write(fh, KG.RAW_BYTES_UNCHECKED)
            with open(pk_file, "rb") as fh:
                pk2 = KG.ProvingKey.read(fh, KG.RAW_BYTES_UNCHECKED, cs)
            pk_bytes = os.path.getsize(pk_file)
            same = all(np.array_equal(x, y) for x, y in zip(
                [pk.l0, pk.l_last, pk.l_active_row, pk.vk.fixed_commitments, pk.vk.permutation_commitments] + pk.fixed_values + pk.fixed_polys + pk.fixed_cosets + pk.permutations + pk.permutation_polys + pk.permutation_cosets,
                [pk2.l0, pk2.l_last, pk2.l_active_row, pk2.vk.fixed_commitments, pk2.vk.permutation_commitments] + pk2.fixed_values + pk2.fixed_polys + pk2.fixed_cosets + pk2.permutations + pk2.permutation_polys + pk2.permutation_cosets))
            assert same, "ProvingKey::read(ProvingKey::write(pk)) differs from pk"
            pk = pk2
            lap("pk_file_round_trip")
            dpk = KG.DeviceProvingKey.from_host(pk, cs)
            del pk, pk2
            lap("pk_upload")

        def from_key(ptr, log_rows):                                               # a working copy of one of the key's columns
            t_ = torch.empty((1 << log_rows, 4), dtype=torch.int64, device=dev)
            KG._copy_device(t_.data_ptr(), ptr, log_rows)
            return t_

        sigma = [from_key(dpk.permutation_values(i), k) for i in range(len(perm_cols))]
        # every set's product column in ONE call (zkhip_permutation_products_device: the sets are chained on the device through z[u]); one
        # read-back of the last set's z[u] says whether the argument closes
        nsets_, npc = cs.num_permutation_sets, len(perm_cols)
        z_all = torch.empty((nsets_, n, 4), dtype=torch.int64, device=dev)
        vptr = (C.c_void_p * npc)(*[pcol(c).data_ptr() for c in range(npc)])
        sptr = (C.c_void_p * npc)(*[sg.data_ptr() for sg in sigma])
        pconsts = [F.fr_encode([v_])[0] for v_ in (beta, gamma, E.DELTA, F.omega_for(k))]
        _lib.check(lib.zkhip_permutation_products_device(vptr, sptr, npc, cs.chunk_len, k, u, *[c_.ctypes.data for c_ in pconsts], z_all.data_ptr(), None))
        z_sets = [z_all[si] for si in range(nsets_)]
        perm_closes = F.fr_decode(z_all[nsets_ - 1, u:u + 1].cpu().numpy().view(np.uint64))[0] == 1
        for z in z_sets:
            z[u + 1:] = rand_fr(n - u - 1)                                         # blinding rows
        lap("permutation_products")

        # ---- lookup argument ---------------------------------------------------------------------------------------------------
        pn, pd = E.lookup_product_programs(1, 1, beta, gamma, theta)
        lookup_cols, lookup_closes = [], True
        for lk in lks:
            pa, ps = rand_fr(n), rand_fr(n)                                         # rows >= u stay random (blinding)
            _lib.check(lib.zkhip_lookup_permute_device(lk.data_ptr(), table.data_ptr(), u, pa.data_ptr(), ps.data_ptr(), None))
            zl = run_prog(pn, [lk, table], k)
            den = run_prog(pd, [pa, ps], k)
            _lib.check(lib.zkhip_fr_grand_product_device(zl.data_ptr(), den.data_ptr(), n, zl.data_ptr(), None))
            lookup_closes = lookup_closes and F.fr_decode(zl[u].cpu().numpy().view(np.uint64))[0] == 1
            zl[u + 1:] = rand_fr(n - u - 1)
            lookup_cols += [zl, pa, ps]
        lap("lookup_permute_and_product")

        # ---- Lagrange -> coefficients, commitments, extended coset ---------------------------------------------------------------
        l0 = torch.zeros((n, 4), dtype=torch.int64, device=dev); l0[0] = ONE
        l_last = torch.zeros((n, 4), dtype=torch.int64, device=dev); l_last[u] = ONE
        l_active = torch.where((rows < u)[:, None], ONE[None, :], torch.zeros_like(ONE)[None, :]).contiguous()
        lagrange = fixed + advice + [l0, l_last, l_active] + sigma + z_sets + lookup_cols
        qc = E.quotient_columns(cs)
        assert len(lagrange) == qc.total
        ncol = len(lagrange)
        coeff = torch.stack(lagrange).contiguous()                                  # [ncol][n][4]
        # the proving key's columns arrive transformed: coefficients (for the evaluations at x) and extended cosets (for the quotient)
        key_polys = [dpk.fixed_poly(i) for i in range(cs.num_fixed)] + [None] * (qc.l0 - qc.advice) + [None, None, None] + [dpk.permutation_poly(i) for i in range(len(perm_cols))]
        key_cosets = ([dpk.fixed_coset(i) for i in range(cs.num_fixed)] + [None] * (qc.l0 - qc.advice) + [dpk.l0(), dpk.l_last(), dpk.l_active_row()]
                      + [dpk.permutation_coset(i) for i in range(len(perm_cols))])
        lap("stack_columns")
        # Columns of the proving key (fixed, l_0 / l_last / l_active, the permutation's sigma polynomials) are transformed once per
        # circuit by keygen and their extended cosets are kept (pk.fixed_cosets, pk.permutation.cosets [DEP]); only the witness-dependent
        # columns (advice, the permutation / lookup products, the permuted lookup pair) are transformed per proof.  Both are done
        # here, in separate batched calls, and timed under separate names: "keygen_*" is not part of the proof.
        pk_ranges = [(qc.fixed, qc.advice), (qc.l0, qc.perm_product)]
        proof_ranges = [(qc.advice, qc.l0), (qc.perm_product, ncol)]

        def ifft_range(lo, hi):
            if hi > lo:
                _lib.check(lib.zkhip_ifft_scaled_batch_device(coeff[lo].data_ptr(), dom.omega_inv.ctypes.data, k, dom.ifft_divisor.ctypes.data, hi - lo, n, None))

        for lo, hi in pk_ranges:                 # l0 / l_last / l_active_row have no stored coefficient form in the key: transformed here, outside the proof time
            for i in range(lo, hi):
                if key_polys[i] is not None:
                    KG._copy_device(coeff[i].data_ptr(), key_polys[i], k)
                else:
                    ifft_range(i, i + 1)
        lap("keygen_lagrange_to_coeff")
        for lo, hi in proof_ranges:
            ifft_range(lo, hi)
        lap("lagrange_to_coeff")
        first_prover_poly = qc.sigma + len(perm_cols)                               # z sets, lookup product, permuted pair
        prod_commit = commit_all(h_g, [coeff[i] for i in range(first_prover_poly, ncol)] + [coeff[qc.advice]])
        a0_coeff_commit = prod_commit.pop()
        lap("commit_products")
        ext = torch.empty((ncol, en, 4), dtype=torch.int64, device=dev)

        def extend_range(lo, hi):
            if hi > lo:
                _lib.check(lib.zkhip_coeff_to_extended_device(coeff[lo].data_ptr(), n, k, ext[lo].data_ptr(), en, ek, hi - lo, dom.extended_omega.ctypes.data,
                                                              dom.g_coset.ctypes.data, None))

        for lo, hi in pk_ranges:
            for i in range(lo, hi):
                KG._copy_device(ext[i].data_ptr(), key_cosets[i], ek)
        lap("keygen_coeff_to_extended")
        for lo, hi in proof_ranges:
            extend_range(lo, hi)
        lap("coeff_to_extended")

        # ---- quotient ---------------------------------------------------------------------------------------------------------------
        if ncol > 96 and ek < 18:
            # hundreds of columns over a few thousand rows: as one program a handful of wavefronts walk thousands of instructions; as a sum of
            # programs over runs of the y-fold's terms (evaluate_h_parts + zkhip_fr_eval_rows_sum_device) they run side by side in one launch
            parts, weights = E.evaluate_h_parts(cs, k, ek, beta, gamma, theta, y, 16)
            h_ext = torch.empty((en, 4), dtype=torch.int64, device=dev)
            E.run_programs_sum_device(parts, weights, [ext[i].data_ptr() for i in range(ncol)], ek, h_ext.data_ptr())
            n_insns = sum(len(p_.insns) for p_ in parts)
            n_regs = 1 + max(max([ins[1] for ins in p_.insns] + [o[1] for ins in p_.insns for o in ins[2:5] if o[0] == E.SRC_REG]) for p_ in parts)
        else:
            prog = E.evaluate_h_program(cs, k, ek, beta, gamma, theta, y)
            h_ext = run_prog(prog, [ext[i] for i in range(ncol)], ek)
            n_insns = len(prog.insns)
            n_regs = 1 + max([ins[1] for ins in prog.insns] + [o[1] for ins in prog.insns for o in ins[2:5] if o[0] == E.SRC_REG])
        tinv = torch.from_numpy(dom.t_evaluations.view(np.int64)).to(dev)
        _lib.check(lib.zkhip_mul_periodic_device(h_ext.data_ptr(), en, tinv.data_ptr(), tinv.shape[0], None))
        lap("evaluate_h")
        h_coeff = torch.empty((en, 4), dtype=torch.int64, device=dev)
        _lib.check(lib.zkhip_extended_to_coeff_device(h_ext.data_ptr(), en, ek, dom.extended_omega_inv.ctypes.data, dom.extended_ifft_divisor.ctypes.data,
                                                      dom.g_coset.ctypes.data, h_coeff.data_ptr(), en, en, 1, None))
        lap("extended_to_coeff")
        h_commit = commit_all(h_g, [h_coeff[i * n:(i + 1) * n] for i in range(3)])
        lap("commit_h")

        # ---- evaluations at x ----------------------------------------------------------------------------------------------------------
        evals = torch.zeros((ncol + 3, 4), dtype=torch.int64, device=dev)
        ptrs = (C.c_void_p * (ncol + 3))(*([coeff[i].data_ptr() for i in range(ncol)] + [h_coeff[i * n:].data_ptr() for i in range(3)]))
        _lib.check(lib.zkhip_fr_eval_polynomial_batch_device(ptrs, ncol + 3, n, F.fr_encode([x])[0].ctypes.data, evals.data_ptr(), None))
        lap("evaluations")

        # ---- multiopen (SHPLONK, the benches' `gen_proof` path): every opened (polynomial, rotation) of the halo2 prover's query plan -----
        from zksnap_circuits_halo2_amd import multiopen as MO

        w_ = F.omega_for(k)
        rot = lambda r: x * pow(w_, r, R) % R
        queries = []
        for i in range(qc.fixed, qc.advice):                       # fixed columns (selectors, constants, table): at x
            queries.append(MO.ProverQuery(rot(0), coeff[i].data_ptr()))
        for i in range(G):                                         # gate advice columns: the vertical gate reads rows 0 .. 3
            for r in range(4):
                queries.append(MO.ProverQuery(rot(r), coeff[qc.advice + i].data_ptr()))
        for j in range(NL):                                        # lookup advice
            queries.append(MO.ProverQuery(rot(0), coeff[qc.advice + G + j].data_ptr()))
        for i in range(qc.sigma, qc.sigma + len(perm_cols)):       # permutation polynomials (proving key)
            queries.append(MO.ProverQuery(rot(0), coeff[i].data_ptr()))
        for si in range(cs.num_permutation_sets):                  # permutation products: x, omega x, and the last usable row for chaining
            zi = coeff[qc.perm_product + si].data_ptr()
            queries += [MO.ProverQuery(rot(0), zi), MO.ProverQuery(rot(1), zi)]
            if si + 1 < cs.num_permutation_sets:
                queries.append(MO.ProverQuery(rot(u), zi))
        for j in range(NL):
            zl_i, pa_i, ps_i = (coeff[qc.lookup + 3 * j + t].data_ptr() for t in range(3))
            queries += [MO.ProverQuery(rot(0), zl_i), MO.ProverQuery(rot(1), zl_i), MO.ProverQuery(rot(0), pa_i), MO.ProverQuery(rot(-1), pa_i),
                        MO.ProverQuery(rot(0), ps_i)]
        for i in range(3):                                         # the quotient's pieces
            queries.append(MO.ProverQuery(rot(0), h_coeff[i * n:].data_ptr()))

        def commit_ptr(ptr):
            out = torch.zeros(12, dtype=torch.int64, device=dev)
            params.commit_device(ptr, n, out.data_ptr())
            return out.cpu().numpy().view(np.uint64)

        y_mo, v_mo, u_mo = (rng.randrange(1, R) for _ in range(3))
        mo_ok = True
        mo_prover = MO.ProverSHPLONK(k, commit_ptr)
        try:
            mo_prover.create_proof(queries, y_mo, v_mo, u_mo)      # raises if L(u) != 0: an evaluation that does not belong to its polynomial
        except ArithmeticError:
            mo_ok = False
        lap("multiopen_shplonk")
        mo_prover.close()

        top_is_zero = not bool(h_coeff[3 * n:].any().item())
        low_nonzero = bool(h_coeff[:3 * n].any().item())
        commit_agrees = affine(adv_commit[0]) == affine(a0_coeff_commit)
        checks = {"permutation_product_closes": perm_closes, "lookup_product_closes": lookup_closes, "quotient_is_a_polynomial": top_is_zero and low_nonzero,
                  "commit_lagrange_equals_commit_coeff": commit_agrees, "multiopen_linearisation_vanishes": mo_ok}
        n_msm = len(adv_commit) + len(prod_commit) + 1 + len(h_commit) + 2
        prove_ms = sum(v for kk, v in t.items() if kk not in ("setup_srs", "witness_columns", "stack_columns", "pk_file_round_trip", "pk_upload") and not kk.startswith("keygen_"))
        n_proof_cols = sum(hi - lo for lo, hi in proof_ranges)
        if verbose:
            print(f"k={k} gate_cols={G} lookups={NL}{' (batched commits)' if batched else ''}: {ncol} columns ({n_proof_cols} witness-dependent, {ncol - n_proof_cols} of the proving key), {n_msm} MSMs of 2^{k}, "
                  f"{n_proof_cols} iNTT 2^{k}, {n_proof_cols} NTT 2^{ek}, 1 iNTT 2^{ek} per proof")
            for name, ms in t.items():
                print(f"  {name:28s} {ms:9.3f} ms")
            print(f"  {'prover steps (no setup/witness)':28s} {prove_ms:9.3f} ms")
            print("  checks:", checks)
        return {"timings_ms": t, "prove_ms": prove_ms, "checks": checks, "columns": ncol, "proof_columns": n_proof_cols, "msms": n_msm, "queries": len(queries),
                "program_insns": n_insns, "program_registers": n_regs,
                "keygen_ms": t.get("keygen_vk", 0.0) + t.get("keygen_pk", 0.0) + t.get("keygen_device", 0.0), "pk_file_bytes": pk_bytes}
    finally:
        torch.cuda.synchronize()
        if dpk is not None:
            dpk.free()
        params.close()


if __name__ == "__main__":
    kk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    gg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ll = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    res = run(kk, gg, lookups=ll)
    sys.exit(0 if all(res["checks"].values()) else 1)
