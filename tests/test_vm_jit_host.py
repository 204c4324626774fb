"""CPU: the run-time compiled row programs (csrc/rowvm_jit.hip).  The generator turns a zkhip_vm_program into straight-line HIP source with the
reductions placed by bounds tracked at generation time; hiprtc cross-compiles it for gfx950 without a device, so both steps are checked here.
On the GPU the same kernels run the row-program tests under ZKHIP_VM_JIT=2 (tests/test_gpu_vm_jit.py).

Replaces the per-row `GraphEvaluator::evaluate` loop of [DEP] halo2-axiom plonk/evaluation.rs (reached from
/root/reference/aggregator/src/wrapper.rs:129); the wrapper's gate shape: /root/reference/aggregator/src/wrapper.rs:792-797."""
import ctypes as C
import re

import pytest

from zksnap_circuits_halo2_amd import _lib, evaluation as E


def source_of(lib, prog, n_columns, log_rows):
    P, keep = prog._marshal()
    n = C.c_size_t(0)
    _lib.check(lib.zkhip_vm_jit_source(C.byref(P), n_columns, log_rows, None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _lib.check(lib.zkhip_vm_jit_source(C.byref(P), n_columns, log_rows, buf, n.value, None))
    del keep
    return buf.value.decode()


def test_wrapper_quotient_program_generates_and_compiles(lib):
    cs = E.halo2_lib_shape(4, 1)
    qc = E.quotient_columns(cs)
    prog = E.evaluate_h_program(cs, 22, 24, 3, 5, 7, 11)
    src = source_of(lib, prog, qc.total, 24)
    body = src[src.index('extern "C" __global__'):]
    n_mul = sum(1 for i in prog.insns if i[0] in (E.OP_MUL, E.OP_SQR, E.OP_MAD))
    assert body.count("fe_mul<Fr, false>(") == n_mul + 1                     # + omega^row
    # lazy additions: reductions only where a product needs them -- a handful, where the interpreter pays one per addition / subtraction
    n_addsub = sum(1 for i in prog.insns if i[0] in (E.OP_ADD, E.OP_SUB, E.OP_MAD, E.OP_NEG, E.OP_DBL))
    assert body.count("condsub2(") + body.count("fe_reduce_soft<Fr>(") <= n_addsub // 4
    assert "A.cols[%d]" % (qc.total - 1) in body and "A.cols[%d]" % qc.total not in body
    assert re.search(r"\(row \+ \d+ull\) & \(A\.rows - 1\)", body)              # rotations are literals, rows a power of two
    assert "struct jit_args { const uint32_t* cols[%d];" % qc.total in src
    code = C.c_size_t(0)
    P, keep = prog._marshal()
    _lib.check(lib.zkhip_vm_jit_compile(C.byref(P), qc.total, 24, C.byref(code)))
    assert code.value > 10000


def test_generator_places_reductions_by_bound(lib):
    """a register that grows by additions is reduced once, before the product that needs it small: 40 additions of a column stay lazy (bound 41),
    the product then takes one quotient-estimate reduction; a subtraction picks the smallest borrow-proof multiple of r above the subtrahend"""
    p = E.RowProgram()
    p.emit(E.OP_MOV, 0, p.column(0))
    for _ in range(40):
        p.emit(E.OP_ADD, 0, E.RowProgram.reg(0), p.column(1))
    p.emit(E.OP_SUB, 1, p.column(0), E.RowProgram.reg(0))                       # subtrahend bound 41 > 15: reduced first (< 2r + 2^233), then 4r - b
    p.emit(E.OP_MUL, 2, E.RowProgram.reg(1), E.RowProgram.reg(0))
    p.result_reg = 2
    src = source_of(lib, p, 2, 10)
    body = src[src.index('extern "C" __global__'):]
    assert body.count("fe_reduce_soft<Fr>(r0)") == 1 and body.count("fe_reduce_soft<Fr>(r1)") + body.count("condsub2(r1)") == 1   # r1 < 1 + 4 = 5: 5 * 2 > 5.29
    assert "Fr::P4_S1" in body and "Fr::P3_S1" not in body and body.count("fe_norm(fe_add(") == 40
    code = C.c_size_t(0)
    P, keep = p._marshal()
    _lib.check(lib.zkhip_vm_jit_compile(C.byref(P), 2, 10, C.byref(code)))


def test_invalid_programs_are_refused_before_generation(lib):
    p = E.RowProgram()
    p.emit(E.OP_MOV, 0, p.column(5))
    P, keep = p._marshal()
    n = C.c_size_t(0)
    assert lib.zkhip_vm_jit_source(C.byref(P), 2, 10, None, 0, C.byref(n)) != 0        # column 5 of 2
    assert b"operand" in lib.zkhip_last_error()
