"""GPU (-m gpu): the row-program tests once more through the run-time compiled kernels (csrc/rowvm_jit.hip).  The switch is read once per
process ($ZKHIP_VM_JIT: 1 = programs of at most 256 instructions over at least 2^18 rows, the default; 2 = every such program whatever the row
count), so the random programs, the add / sub chains at the bounds, the evaluate_h formula checks and the accumulate test of
tests/test_gpu_rows.py run in a child process with ZKHIP_VM_JIT=2 -- the same expectations (the oracle's interpreter, the reference's formulas)
met by generated straight-line code -- and a timing line compares the two executions of the wrapper's quotient at 2^22 rows."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_row_program_suite_through_compiled_kernels():
    env = dict(os.environ, ZKHIP_VM_JIT="2", ZKHIP_VM_JIT_LOG="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_rows.py"), "-x", "-q", "-m", "gpu", "-k",
                          "random_vs_interpreter or chains_at_the_bounds or evaluate_h_matches or linear_combination or golden_prover_steps"],
                         capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "compilation failed" not in res.stderr, res.stderr[-2000:]
    assert " passed" in res.stdout


@pytest.mark.gpu
def test_compiled_and_interpreted_quotient_agree_bit_for_bit():
    """the wrapper-shaped quotient program over 2^18 rows of random columns: ZKHIP_VM_JIT=0 and the default must write the same bytes"""
    code = r'''
import sys, hashlib
sys.path.insert(0, %r)
import random, numpy as np, torch
from zksnap_circuits_halo2_amd import _lib, evaluation as E, fields as F
cs = E.halo2_lib_shape(4, 1)
qc = E.quotient_columns(cs)
rng = random.Random(5)
beta, gamma, theta, y = (rng.randrange(F.R_MOD) for _ in range(4))
ek = 18
prog = E.evaluate_h_program(cs, ek - 2, ek, beta, gamma, theta, y)
torch.manual_seed(7)
cols = []
for _ in range(qc.total):
    t = torch.randint(0, 1 << 62, (1 << ek, 4), dtype=torch.int64, device="cuda")
    t[:, 3] &= (1 << 61) - 1
    cols.append(t)
out = torch.zeros((1 << ek) * 4, dtype=torch.int64, device="cuda")
prog.run_device([t.data_ptr() for t in cols], ek, out.data_ptr())
torch.cuda.synchronize()
print("DIGEST", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
''' % ROOT
    digests = []
    for mode in ("0", "1"):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, ZKHIP_VM_JIT=mode, ZKHIP_VM_JIT_LOG="1"))
        assert res.returncode == 0 and "compilation failed" not in res.stderr, res.stdout + res.stderr[-2000:]
        digests.append([l for l in res.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1]
