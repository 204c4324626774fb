"""GPU: the Rust shim's call sequence (rust-shim/) issued from C99 -- tests/cpp/shim_sequence.c -- and its commitments against the oracle.

The reference's prover reaches the library only through the patched halo2-axiom crate (`create_proof`,
/root/reference/aggregator/src/wrapper.rs:129; `gen_pk`, wrapper.rs:106-109); no Rust toolchain exists here, so the sequence the shim
issues over the life of two `ParamsKZG` objects (self-test, register, commits on `&g[..n]` sub-slices, transforms, Drop = unregister, a
second SRS at the same addresses, register again) is replayed through the same C ABI by a strict-C99 program.  The program checks the
registered path against the unregistered one itself; this test also checks the printed commitments against the structured identity
MSM(a, (t0 + i d) G) = [sum a_i (t0 + i d)] G computed by the C oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import fields as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK = (1 << 64) - 1


def xorshift_scalars(n, seed):
    """the generator of shim_sequence.c (xorshift64*), as (n, 4) uint64 Montgomery words"""
    out = np.zeros((n, 4), dtype=np.uint64)
    s = seed
    for i in range(n):
        for j in range(4):
            s ^= s >> 12
            s ^= (s << 25) & MASK
            s ^= s >> 27
            v = (s * 0x2545F4914F6CDD1D) & MASK
            out[i, j] = v & 0x0FFFFFFFFFFFFFFF if j == 3 else v
    return out


@pytest.mark.gpu
def test_shim_call_sequence_from_c99(tmp_path, cref):
    exe = tmp_path / "shim_sequence"
    lib_dir = os.path.join(ROOT, "zksnap_circuits_halo2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_sequence.c"), "-o", str(exe), "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "shim sequence OK" in res.stdout
    commits = {}
    for line in res.stdout.splitlines():
        if line.startswith("commit "):
            _, tag, n, *words = line.split()
            w = [int(x, 16) for x in words]
            commits[tag] = (int(n), np.array(w, dtype=np.uint64))
    assert set(commits) == {"g_unregistered", "g_lagrange_unregistered", "g_registered", "g_lagrange_registered", "g2_registered"}
    # the constants of the C file: omega_12, omega_14, their inverses, 2^-12, 2^-14, zeta -- re-derived
    src = open(os.path.join(ROOT, "tests", "cpp", "shim_sequence.c")).read()
    const = lambda name: [int(x, 16) for x in re.search(r"\b" + name + r"\[4\] = \{([^}]*)\}", src).group(1).replace("ULL", "").split(",")]
    enc = lambda v: [int(x) for x in F.fr_encode([v])[0]]
    w12, w14 = F.omega_for(12), F.omega_for(14)
    assert const("OMEGA") == enc(w12) and const("OMEGA_INV") == enc(pow(w12, -1, F.R_MOD)) and const("N_INV") == enc(pow(1 << 12, -1, F.R_MOD))
    assert const("EXT_OMEGA") == enc(w14) and const("EXT_OMEGA_INV") == enc(pow(w14, -1, F.R_MOD)) and const("EXT_DIV") == enc(pow(1 << 14, -1, F.R_MOD))
    assert const("ZETA") == enc(F.ZETA) and const("FR_ONE") == enc(1)
    # commitments against the structured identity (walk parameters were passed as Montgomery words (t, 0, 0, 0))
    n_full = 1 << 14
    scalars = xorshift_scalars(n_full, 0x5A4B534E41500001)
    word = lambda t: F.fr_decode(np.array([[t, 0, 0, 0]], dtype=np.uint64))[0]
    def expect(n, t0, d):
        k = cref.expected_scalar(np.ascontiguousarray(scalars[:n]), word(t0), word(d))
        return F.g1_decode_jacobian(cref.scalar_mul(k, cref.generator()))
    def got(tag):
        n, aff = commits[tag]
        return n, O.affine_from_limbs([int(x) for x in aff])
    for tag, (t0, d) in {"g_unregistered": (5, 7), "g_registered": (5, 7), "g_lagrange_unregistered": (1000003, 11), "g_lagrange_registered": (1000003, 11),
                         "g2_registered": (900001, 13)}.items():
        n, pt = got(tag)
        assert pt == expect(n, t0, d), tag
