"""GPU: the call sequence of rust-shim/prover_patch.rs for one proof, issued from C99 (tests/cpp/prover_sequence.c) three ways over the same
witness -- host buffers call by call (what an unmodified create_proof makes through the shim of rounds 1-4), host buffers with one call per
phase (the patch's mode (a)), device-resident handles (mode (b)) -- every commitment, evaluation and quotient coefficient of the last two
compared with the first by the program itself.

Workload: the state-transition shape at its bench k = 15 (/root/reference/aggregator/benches/state_transition_circuit.rs:22,84: 3 gate
columns + 1 lookup column, /root/reference/aggregator/benches/wrapper_circuit.rs:41-48), reached through create_proof
(/root/reference/aggregator/src/wrapper.rs:129).  The row programs come from evaluation.export_prover_programs: the `GraphEvaluator` ->
zkhip_vm_insn translation that the patched [DEP] plonk/evaluation.rs performs on the Rust side."""
import os
import re
import subprocess

import pytest

from zksnap_circuits_halo2_amd import evaluation as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = tmp_path / "prover_sequence"
    lib_dir = os.path.join(ROOT, "zksnap_circuits_halo2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "prover_sequence.c"), "-o", str(exe), "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("k,gate_cols", [(15, 3), (11, 1), (13, 6), (10, 40)])      # (10, 40): a 600-instruction quotient program, run as a sum of parts by sequence D
def test_prover_patch_call_sequence_from_c99(tmp_path, k, gate_cols):
    exe = build(tmp_path)
    rec = tmp_path / "programs.bin"
    rec.write_bytes(E.export_prover_programs(k, gate_cols, 1, seed=k))
    res = subprocess.run([str(exe), str(rec)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "prover sequence OK" in res.stdout
    m = re.search(r"commitments compared: (\d+), evaluations compared: (\d+), quotient coefficients compared: (\d+)", res.stdout)
    nadv = gate_cols + 1
    nsets = -(-(nadv + 1) // 2)
    assert m and int(m.group(1)) == nadv + 2 + nsets + 1 + 3 and int(m.group(2)) == nadv + nsets + 3 + 3 and int(m.group(3)) == 3 << k
    assert re.search(r"sequence_ms host_call_by_call=[\d.]+ host_one_call_per_phase=[\d.]+ device_resident=[\d.]+", res.stdout)
    nq = (gate_cols + 2) + 4 * gate_cols + 1 + (nadv + 1) + (3 * nsets - 1) + 5 + 3
    assert f"multiopen on the device: {nq} queries, SHPLONK H and H', GWC 6 witnesses, a wrong evaluation refused" in res.stdout


def test_exported_programs_record_is_well_formed():
    """(CPU) the record's header and the count of programs: magic, shape words, 20 constants (domain, x, omega, delta, beta, gamma, 6 opening points, 3 multi-open challenges), t_evaluations, then
    to_mont + 2 * sets + 2 + 1 programs, consumed to the last byte"""
    import struct

    k, G = 9, 2
    b = E.export_prover_programs(k, G, 1)
    assert b[:4] == b"ZKPS" and struct.unpack_from("<I", b, 4)[0] == 3
    kk, ek, g, nl, nperm, nsets, chunk, blind = struct.unpack_from("<8I", b, 8)
    assert (kk, ek, g, nl, nperm, nsets, chunk, blind) == (k, k + 2, G, 1, G + 2, 2, 2, 5)
    off = 8 + 32 + 28 + 20 * 32
    period = struct.unpack_from("<I", b, off)[0]
    off += 4 + period * 32
    assert period == 4
    nprog = 0
    assert struct.unpack_from("<I", b, len(b) - 4)[0] == 0           # the count of quotient parts: none for a 100-instruction program
    b = b[:-4]
    while off < len(b):
        n_insns = struct.unpack_from("<I", b, off)[0]; off += 4 + 16 * n_insns
        n_const = struct.unpack_from("<I", b, off)[0]; off += 4 + 32 * n_const
        n_rot = struct.unpack_from("<I", b, off)[0]; off += 4 + 4 * n_rot
        _, _, has_omega = struct.unpack_from("<iII", b, off); off += 12 + (32 if has_omega else 0)
        assert n_insns >= 1
        nprog += 1
    assert off == len(b) and nprog == 1 + 2 * nsets + 2 + 1
