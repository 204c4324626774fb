"""CPU (`-m "not gpu"`): the device arithmetic compiled for the host, and the C oracle, both under ASan + UBSan.

* tests/cpp/fp29_host_check.cpp includes csrc/fp29.hpp + csrc/ec.hpp (the code the kernels run) and checks the lazy radix-2^29
  field, the external-format conversions and the XYZZ formulas -- values *and* the documented magnitude bounds -- against an
  independent 4x64 Montgomery big-integer implementation.
* tests/cpp/oracle_sanitize.c drives oracle/cpu_ref.c (threads, both FFT paths, row a7).
GPU AddressSanitizer is not available on this pool, so this is where the sanitizers run."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1"]


def _run(cmd, **kw):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600, **kw)


def test_device_field_and_curve_arithmetic_on_host_with_sanitizers(tmp_path):
    exe = tmp_path / "fp29_host_check"
    build = _run(["g++", "-std=c++17", *SAN, "-I", os.path.join(ROOT, "zksnap_circuits_halo2_amd", "csrc"),
                  os.path.join(ROOT, "tests", "cpp", "fp29_host_check.cpp"), "-o", str(exe)])
    assert build.returncode == 0, build.stdout
    run = _run([str(exe)])
    assert run.returncode == 0 and "host check OK" in run.stdout, run.stdout


def test_c_oracle_with_sanitizers(tmp_path):
    exe = tmp_path / "oracle_sanitize"
    build = _run(["gcc", *SAN, "-pthread", os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "cpu_ref.c"),
                  "-o", str(exe), "-lm"])
    assert build.returncode == 0, build.stdout
    run = _run([str(exe)])
    assert run.returncode == 0 and "oracle sanitize OK" in run.stdout, run.stdout


def test_cpp_mirror_host_arithmetic_with_sanitizers(tmp_path):
    """the host-only arithmetic of include/zkhip.hpp (both fields in 4 x 64 Montgomery form, Fq2 square roots, G2 compression, the SHPLONK
    prover's interpolation) against Python integers, under ASan + UBSan; nothing here touches the GPU library"""
    import random
    import struct

    import numpy as np

    from zksnap_circuits_halo2_amd import fields as F, multiopen as M, srs

    exe = tmp_path / "mirror_host_check"
    build = _run(["g++", "-std=c++17", *SAN, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_host_check.cpp"), "-o", str(exe)])
    assert build.returncode == 0, build.stdout
    rng = random.Random(12)
    pts = [srs.G2_GENERATOR, srs.g2_mul(0xC0FFEE), srs.g2_mul(rng.randrange(F.R_MOD)), srs.g2_mul(F.R_MOD - 1), None]
    xs = [rng.randrange(F.R_MOD) for _ in range(4)]
    ys = [rng.randrange(F.R_MOD) for _ in range(4)]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<I", len(pts)))
        for P in pts:
            f.write(srs.g2_encode(P).tobytes())
        f.write(struct.pack("<I", len(xs)))
        for x, y in zip(xs, ys):
            f.write(F.fr_encode([x, y]).tobytes())
    run = _run([str(exe), str(fin), str(fout)])
    assert run.returncode == 0 and "mirror host check done" in run.stdout, run.stdout
    out = fout.read_bytes()
    m, off = len(pts), 0
    for layout in (0, 1):
        for P in pts:
            assert out[off:off + 64] == srs.g2_compress(P, layout), (layout, P)
            off += 64
    for layout in (0, 1):
        for P in pts:
            assert np.array_equal(np.frombuffer(out[off:off + 128], dtype=np.uint64), srs.g2_encode(P))
            off += 128
    coeffs = F.fr_decode(np.frombuffer(out[off:off + 32 * len(xs)], dtype=np.uint64).reshape(-1, 4))
    off += 32 * len(xs)
    assert coeffs == M._interpolate(xs, ys)
    flags, = struct.unpack("<Q", out[off:off + 8])
    assert flags == 0b11111, bin(flags)          # bit 4: g2_is_valid accepts the points and rejects off-curve / non-canonical ones
