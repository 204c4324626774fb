"""CPU: the C-ABI library loads and exports every symbol include/zkhip.h declares; host-side logic (formats, domain
constants, sharding) is right; without a GPU every compute entry point fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from oracle import bn254 as O
from zksnap_circuits_halo2_amd import _lib, fields as F
from zksnap_circuits_halo2_amd.domain import EvaluationDomain
from zksnap_circuits_halo2_amd.multi_gpu import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol(lib):
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zkhip.h but not exported by libzkhip.so"
    assert names == _lib.exported_symbols(), "ctypes signature table out of sync with include/zkhip.h"


def _no_gpu():
    try:
        import torch

        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful without a GPU")
def test_compute_fails_loudly_without_gpu(lib):
    out = np.zeros(12, dtype=np.uint64)
    sc = np.zeros((4, 4), dtype=np.uint64)
    bs = np.zeros((4, 8), dtype=np.uint64)
    assert lib.zkhip_msm_g1(sc.ctypes.data, bs.ctypes.data, 4, out.ctypes.data) == -2   # ZKHIP_ENODEV
    assert b"HIP device" in lib.zkhip_last_error()
    assert lib.zkhip_ntt_fr(sc.ctypes.data, sc.ctypes.data, 2) == -2
    with pytest.raises(_lib.ZkhipError):
        import zksnap_circuits_halo2_amd as Z

        Z.best_multiexp(sc, bs)


def test_argument_validation_has_no_side_effects(lib):
    assert lib.zkhip_init(None, -1) == -1 and lib.zkhip_init(None, 65) == -1   # ZKHIP_EINVAL before any device is touched
    two = (__import__("ctypes").c_int * 2)(0, 0)
    assert lib.zkhip_init(two, 2) in (-1, -2)   # a device named twice (or, on a box without a GPU, no device at all)
    assert lib.zkhip_msm_window_bits(1 << 20) == 16
    assert 2 <= lib.zkhip_msm_window_bits(1) <= 16


def test_format_round_trips():
    g = O.SplitMix64(3)
    vals = [0, 1, O.R_MOD - 1] + [g.fr() for _ in range(20)]
    enc = F.fr_encode(vals)
    assert enc.shape == (23, 4) and F.fr_decode(enc) == vals
    assert enc[1].tolist() == O.fr_to_limbs(1)
    pts = [None, O.G1_GEN, O.scalar_mul(7, O.G1_GEN)]
    e = F.g1_encode(pts)
    assert [O.affine_from_limbs([int(x) for x in r]) for r in e] == pts
    jac = np.array(O.limbs4(O.to_mont(4, O.Q_MOD)) + O.limbs4(O.to_mont(16, O.Q_MOD)) + O.limbs4(O.to_mont(2, O.Q_MOD)), dtype=np.uint64)
    assert F.g1_decode_jacobian(jac) == (1, 2)      # (4/2^2, 16/2^3)
    assert F.g1_decode_jacobian(np.zeros(12, dtype=np.uint64)) is None


@pytest.mark.parametrize("j,k", [(4, 13), (4, 15), (4, 22), (3, 5), (9, 4)])
def test_domain_constants_match_oracle(j, k):
    d, o = EvaluationDomain(j, k), O.EvaluationDomain(j, k)
    assert d.extended_k == o.extended_k and d.quotient_poly_degree == o.quotient_poly_degree
    dec = lambda v: F.fr_decode(v)[0]
    assert dec(d.omega) == o.omega and dec(d.omega_inv) == o.omega_inv
    assert dec(d.extended_omega) == o.extended_omega and dec(d.extended_omega_inv) == o.extended_omega_inv
    assert dec(d.ifft_divisor) == o.ifft_divisor and dec(d.extended_ifft_divisor) == o.extended_ifft_divisor
    assert F.fr_decode(d.t_evaluations) == o.t_evaluations
    assert pow(o.omega, d.n, O.R_MOD) == 1 and pow(o.omega, d.n // 2, O.R_MOD) != 1


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_header_is_plain_c99_and_links(tmp_path):
    """include/zkhip.h is what a cgo / Rust-bindgen / JNI binding would consume: it must compile as strict C99, and a C program must
    link against libzkhip.so with it (no C++ in the interface)"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "zkhip.h"\n'
                   "int main(void) {\n"
                   "  zkhip_vm_insn insn; zkhip_vm_program prog; (void)insn; (void)prog;\n"
                   "  /* no GPU here: the call must fail cleanly with ZKHIP_ENODEV and leave a message */\n"
                   "  unsigned long long a[4] = {0, 0, 0, 0}, w[4] = {1, 0, 0, 0};\n"
                   "  int rc = zkhip_ntt_fr((uint64_t *)a, (const uint64_t *)w, 0);\n"
                   "  return (rc == ZKHIP_ENODEV || rc == ZKHIP_OK) && zkhip_last_error() != 0 ? 0 : 1;\n"
                   "}\n")
    exe = tmp_path / "abi"
    lib_dir = os.path.join(root, "zksnap_circuits_halo2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
    assert subprocess.call([str(exe)]) == 0


def test_generated_mac_blocks_header_is_current(tmp_path):
    """zksnap_circuits_halo2_amd/csrc/mac_blocks.hpp is generated by tools/gen_mac_blocks.py: the committed file must be what the
    generator writes (the asm blocks are the shape of the bucket-accumulation kernel's field products)"""
    import importlib.util
    import shutil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(root, "zksnap_circuits_halo2_amd", "csrc", "mac_blocks.hpp")).read()
    # run the generator on a copy of the tree layout it expects
    tools = tmp_path / "tools"; csrc = tmp_path / "zksnap_circuits_halo2_amd" / "csrc"
    tools.mkdir(); csrc.mkdir(parents=True)
    shutil.copy(os.path.join(root, "tools", "gen_mac_blocks.py"), tools / "gen_mac_blocks.py")
    spec = importlib.util.spec_from_file_location("gen_mac_blocks", tools / "gen_mac_blocks.py")
    spec.loader.exec_module(importlib.util.module_from_spec(spec))
    assert (csrc / "mac_blocks.hpp").read_text() == committed
    assert committed.count("v_mad_u64_u32") >= 2 * sum(range(1, 10))


def test_domain_scalar_helpers():
    """rotate_omega / l_i_range of the EvaluationDomain mirror against their definitions (host big-int arithmetic only)"""
    d = EvaluationDomain(4, 5)
    n, w = d.n, O.omega_for(5)
    g = O.SplitMix64(5)
    x = g.fr()
    assert d.rotate_omega(x, 3) == x * pow(w, 3, O.R_MOD) % O.R_MOD
    assert d.rotate_omega(x, -2) == x * pow(w, -2, O.R_MOD) % O.R_MOD
    xn = pow(x, n, O.R_MOD)
    rots = list(range(-3, 4))
    got = d.l_i_range(x, xn, rots)
    for r, v in zip(rots, got):      # l_i(x) = prod_{j != i} (x - w^j) / (w^i - w^j)
        i = r % n
        num = den = 1
        for j in range(n):
            if j != i:
                num = num * (x - pow(w, j, O.R_MOD)) % O.R_MOD
                den = den * (pow(w, i, O.R_MOD) - pow(w, j, O.R_MOD)) % O.R_MOD
        assert v == num * pow(den, -1, O.R_MOD) % O.R_MOD, r
    assert d.l_i_range(pow(w, 2, O.R_MOD), 1, [2, 3]) == [1, 0]
