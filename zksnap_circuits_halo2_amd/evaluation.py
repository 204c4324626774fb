"""`halo2_proofs::plonk::evaluation` mirror [DEP halo2-axiom plonk/evaluation.rs], reached from create_proof
(/root/reference/aggregator/src/wrapper.rs:129): the quotient numerator h(X)·(X^n − 1) is evaluated row by row over the
extended coset.  The reference compiles every gate / lookup `Expression` into a `GraphEvaluator` (a list of
`Calculation`s over `ValueSource`s, hash-consed) and interprets it per row on the CPU, then adds the permutation and lookup
terms with hand-written loops.  Here the same graph -- including the permutation and lookup terms -- is compiled once into a
row program for `zkhip_fr_eval_rows` (include/zkhip.h) and run as one fused GPU pass, one thread per row.

Host logic only: graph construction, register allocation, argument marshalling.  All field arithmetic on columns happens in
libzkhip.so; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import fields as F

# operand kinds / opcodes of include/zkhip.h
SRC_CONST, SRC_REG, SRC_COLUMN, SRC_PREV, SRC_ROWPOW = 0, 1, 2, 3, 4
OP_MOV, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_DBL, OP_SQR, OP_MAD = range(8)
_N_OPERANDS = {OP_MOV: 1, OP_ADD: 2, OP_SUB: 2, OP_MUL: 2, OP_NEG: 1, OP_DBL: 1, OP_SQR: 1, OP_MAD: 3}


# ---------------------------------------------------------------------------------------------------
# Expression: the reference's `plonk::Expression<F>` as a tiny AST
# ---------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Expr:
    kind: str                      # constant | fixed | advice | instance | challenge | neg | sum | product | scaled
    a: object = None
    b: object = None

    def __add__(self, o): return Expr("sum", self, o)
    def __sub__(self, o): return Expr("sum", self, Expr("neg", o))
    def __mul__(self, o): return Expr("product", self, o) if isinstance(o, Expr) else Expr("scaled", self, int(o))
    def __neg__(self): return Expr("neg", self)


def Constant(v: int) -> Expr: return Expr("constant", int(v) % F.R_MOD)
def Fixed(col: int, rot: int = 0) -> Expr: return Expr("fixed", col, rot)
def Advice(col: int, rot: int = 0) -> Expr: return Expr("advice", col, rot)
def Instance(col: int, rot: int = 0) -> Expr: return Expr("instance", col, rot)
def Challenge(i: int) -> Expr: return Expr("challenge", i)


@dataclass
class Lookup:                      # `lookup::Argument`
    input_expressions: List[Expr]
    table_expressions: List[Expr]


@dataclass
class ConstraintSystem:            # the parts of `plonk::ConstraintSystem` evaluate_h reads
    num_fixed: int
    num_advice: int
    num_instance: int = 0
    gates: List[List[Expr]] = field(default_factory=list)                  # per gate: its polynomials
    lookups: List[Lookup] = field(default_factory=list)
    permutation_columns: List[Tuple[str, int]] = field(default_factory=list)   # ("advice" | "fixed" | "instance", index)
    blinding_factors: int = 5
    degree: int = 4                                                         # cs.degree(); chunk_len = degree - 2

    @property
    def chunk_len(self) -> int: return self.degree - 2
    @property
    def num_permutation_sets(self) -> int:
        return (len(self.permutation_columns) + self.chunk_len - 1) // self.chunk_len if self.permutation_columns else 0


# ---------------------------------------------------------------------------------------------------
# Graph: hash-consed calculations (the reference's GraphEvaluator::add_calculation / add_constant / add_rotation)
# ---------------------------------------------------------------------------------------------------
Node = Tuple  # ("const", value) | ("col", column, rot) | ("prev",) | ("rowpow",) | ("op", opcode, a_id, b_id, c_id)


class Graph:
    def __init__(self):
        self.nodes: List[Node] = []
        self._index: Dict[Node, int] = {}

    def _add(self, node: Node) -> int:
        i = self._index.get(node)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(node)
            self._index[node] = i
        return i

    def const(self, v: int) -> int: return self._add(("const", int(v) % F.R_MOD))
    def col(self, column: int, rot: int = 0) -> int: return self._add(("col", int(column), int(rot)))
    def prev(self) -> int: return self._add(("prev",))
    def rowpow(self) -> int: return self._add(("rowpow",))

    def op(self, opcode: int, a: int, b: int = -1, c: int = -1) -> int:
        if opcode in (OP_ADD, OP_MUL) and b < a:      # commutative: canonical operand order (as the reference does)
            a, b = b, a
        return self._add(("op", opcode, a, b, c))

    def add(self, a, b): return self.op(OP_ADD, a, b)
    def sub(self, a, b): return self.op(OP_SUB, a, b)
    def mul(self, a, b): return self.op(OP_SQR, a) if a == b else self.op(OP_MUL, a, b)
    def neg(self, a): return self.op(OP_NEG, a)
    def dbl(self, a): return self.op(OP_DBL, a)
    def mad(self, a, b, c):                                                   # a * b + c
        if self.nodes[a] == ("const", 0) or self.nodes[b] == ("const", 0):
            return c                                                          # 0 * b + c (the first Horner step of the reference)
        return self._add(("op", OP_MAD, a, b, c))

    def horner(self, start: int, parts: Sequence[int], factor: int) -> int:
        """`Calculation::Horner(start, parts, factor)`: value = start; value = value * factor + part for each part."""
        v = start
        for p in parts:
            v = self.mad(v, factor, p)
        return v


# ---------------------------------------------------------------------------------------------------
# RowProgram: the marshalled form zkhip_fr_eval_rows takes
# ---------------------------------------------------------------------------------------------------
class RowProgram:
    def __init__(self, rot_scale: int = 1, omega: Optional[int] = None):
        self.insns: List[Tuple[int, int, Tuple[int, int, int], Tuple[int, int, int], Tuple[int, int, int]]] = []
        self.constants: List[int] = []
        self._const_index: Dict[int, int] = {}
        self.rotations: List[int] = []
        self._rot_index: Dict[int, int] = {}
        self.rot_scale = rot_scale
        self.omega = omega
        self.result_reg = 0
        self.n_columns = 0

    # operands are (kind, index, rot_slot)
    def constant(self, v: int):
        v = int(v) % F.R_MOD
        if v not in self._const_index:
            self._const_index[v] = len(self.constants)
            self.constants.append(v)
        return (SRC_CONST, self._const_index[v], 0)

    def column(self, column: int, rot: int = 0):
        if rot not in self._rot_index:
            self._rot_index[rot] = len(self.rotations)
            self.rotations.append(rot)
        self.n_columns = max(self.n_columns, column + 1)
        return (SRC_COLUMN, column, self._rot_index[rot])

    @staticmethod
    def reg(i: int): return (SRC_REG, i, 0)
    PREV = (SRC_PREV, 0, 0)
    ROWPOW = (SRC_ROWPOW, 0, 0)

    def emit(self, op: int, dst: int, a, b=None, c=None):
        z = (SRC_CONST, 0, 0)
        self.insns.append((op, dst, a, b or z, c or z))

    # ---- marshalling -------------------------------------------------------------------------------
    def _marshal(self):
        n = len(self.insns)
        arr = (_lib.VmInsn * n)()
        for i, (op, dst, a, b, c) in enumerate(self.insns):
            arr[i].op, arr[i].dst = op, dst
            for name, o in (("a", a), ("b", b), ("c", c)):
                f = getattr(arr[i], name)
                f.kind, f.index, f.rot = o[0], o[1], o[2]
        consts = F.fr_encode(self.constants) if self.constants else np.zeros((1, 4), dtype=np.uint64)
        rots = (C.c_int32 * max(len(self.rotations), 1))(*self.rotations)
        omega = F.fr_encode([self.omega])[0] if self.omega is not None else None
        prog = _lib.VmProgram()
        prog.insns, prog.n_insns = arr, n
        prog.constants, prog.n_constants = consts.ctypes.data, len(self.constants)
        prog.rotations, prog.n_rotations = rots, len(self.rotations)
        prog.rot_scale, prog.result_reg = self.rot_scale, self.result_reg
        prog.omega = omega.ctypes.data if omega is not None else None
        keep = (arr, consts, rots, omega)      # keep the buffers alive for the duration of the call
        return prog, keep

    def to_bytes(self) -> bytes:
        """The program as the flat little-endian record a non-Python host reads back into a `zkhip_vm_program` (include/zkhip.h):
        u32 n_insns, n_insns x 16-byte zkhip_vm_insn, u32 n_constants, n_constants x 4 u64 (Montgomery words), u32 n_rotations,
        n_rotations x i32, i32 rot_scale, u32 result_reg, u32 has_omega, 4 u64 omega when has_omega.  This is what a patched
        [DEP] plonk/evaluation.rs produces from its `GraphEvaluator` (rust-shim/prover_patch.rs); tests/cpp/prover_sequence.c loads it."""
        import struct

        out = [struct.pack("<I", len(self.insns))]
        for op, dst, a, b, c in self.insns:
            out.append(struct.pack("<BBH", op, dst, 0))
            for o in (a, b, c):
                out.append(struct.pack("<BBH", o[0], o[2], o[1]))          # kind, rot slot, index (zkhip_vm_operand)
        out.append(struct.pack("<I", len(self.constants)))
        if self.constants:
            out.append(F.fr_encode(self.constants).astype("<u8").tobytes())
        out.append(struct.pack("<I", len(self.rotations)))
        out.append(struct.pack("<%di" % len(self.rotations), *self.rotations))
        out.append(struct.pack("<iII", self.rot_scale, self.result_reg, 1 if self.omega is not None else 0))
        if self.omega is not None:
            out.append(F.fr_encode([self.omega])[0].astype("<u8").tobytes())
        return b"".join(out)

    def run(self, columns: Sequence[np.ndarray], log_rows: int, out: Optional[np.ndarray] = None, accumulate: bool = False) -> np.ndarray:
        """Host columns ((2^log_rows, 4) uint64 each) -> (2^log_rows, 4) uint64."""
        rows = 1 << log_rows
        cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(rows, 4) for c in columns]
        assert len(cols) >= self.n_columns, "program reads more columns than were passed"
        if out is None:
            assert not accumulate
            out = np.zeros((rows, 4), dtype=np.uint64)
        assert out.dtype == np.uint64 and out.flags.c_contiguous and out.size == 4 * rows
        ptrs = (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
        prog, keep = self._marshal()
        _lib.check(_lib.load().zkhip_fr_eval_rows(C.byref(prog), ptrs, len(cols), log_rows, int(accumulate), out.ctypes.data))
        del keep
        return out

    def run_device(self, d_columns: Sequence[int], log_rows: int, d_out: int, accumulate: bool = False, stream: int = 0) -> None:
        """Device-resident columns: `d_columns` are device addresses (e.g. torch `tensor.data_ptr()`)."""
        assert len(d_columns) >= self.n_columns, "program reads more columns than were passed"
        ptrs = (C.c_void_p * max(len(d_columns), 1))(*d_columns)
        prog, keep = self._marshal()
        _lib.check(_lib.load().zkhip_fr_eval_rows_device(C.byref(prog), ptrs, len(d_columns), log_rows, int(accumulate), d_out, stream))
        del keep


def compile_graph(g: Graph, result: int, rot_scale: int = 1, omega: Optional[int] = None) -> RowProgram:
    """Lower the calculations `result` depends on to a RowProgram: creation order is a valid schedule (operands precede their
    users), leaves become instruction operands, and registers are reused after an intermediate's last use."""
    needed = set()
    stack = [result]
    while stack:
        i = stack.pop()
        if i in needed:
            continue
        needed.add(i)
        n = g.nodes[i]
        if n[0] == "op":
            stack.extend(x for x in n[2:5] if x >= 0)
    order = [i for i in sorted(needed) if g.nodes[i][0] == "op"]
    last_use: Dict[int, int] = {}
    for pos, i in enumerate(order):
        for x in g.nodes[i][2:5]:
            if x >= 0:
                last_use[x] = pos
    last_use[result] = len(order)
    prog = RowProgram(rot_scale=rot_scale, omega=omega)
    free = list(range(_lib.VM_REGS - 1, -1, -1))
    reg_of: Dict[int, int] = {}

    def operand(x: int):
        n = g.nodes[x]
        if n[0] == "const": return prog.constant(n[1])
        if n[0] == "col": return prog.column(n[1], n[2])
        if n[0] == "prev": return RowProgram.PREV
        if n[0] == "rowpow":
            assert omega is not None, "graph uses omega^row but no omega was given"
            return RowProgram.ROWPOW
        return RowProgram.reg(reg_of[x])

    for pos, i in enumerate(order):
        _, opcode, a, b, c = g.nodes[i]
        ops = [operand(x) for x in (a, b, c)[: _N_OPERANDS[opcode]]]
        # operands whose last use is this instruction release their register before the destination is chosen
        for x in {a, b, c}:
            if x >= 0 and g.nodes[x][0] == "op" and last_use[x] == pos:
                free.append(reg_of.pop(x))
        if not free:
            raise ValueError(f"row program needs more than {_lib.VM_REGS} live registers")
        reg_of[i] = free.pop()
        prog.emit(opcode, reg_of[i], *ops)
    if g.nodes[result][0] != "op":       # the result is a leaf: materialise it
        prog.emit(OP_MOV, 0, operand(result))
        prog.result_reg = 0
    else:
        prog.result_reg = reg_of[result]
    return prog


# ---------------------------------------------------------------------------------------------------
# evaluate_h: column layout + graph of the whole quotient numerator
# ---------------------------------------------------------------------------------------------------
@dataclass
class QuotientColumns:
    """Column order `evaluate_h_program` expects (all are evaluations over the extended coset, 2^extended_k rows each)."""
    fixed: int
    advice: int
    instance: int
    l0: int
    l_last: int
    l_active_row: int
    sigma: int            # first of the permutation cosets (pk.permutation.cosets), one per permutation column
    perm_product: int     # first of the permutation product cosets, one per set
    lookup: int           # first lookup triple: product, permuted_input, permuted_table per lookup
    total: int


def quotient_columns(cs: ConstraintSystem) -> QuotientColumns:
    o = 0
    fixed = o; o += cs.num_fixed
    advice = o; o += cs.num_advice
    instance = o; o += cs.num_instance
    l0 = o; l_last = o + 1; l_active = o + 2; o += 3
    sigma = o; o += len(cs.permutation_columns)
    perm = o; o += cs.num_permutation_sets
    lookup = o; o += 3 * len(cs.lookups)
    return QuotientColumns(fixed, advice, instance, l0, l_last, l_active, sigma, perm, lookup, o)


def _add_expression(g: Graph, cs: ConstraintSystem, qc: QuotientColumns, e: Expr, challenges: Sequence[int]) -> int:
    k = e.kind
    if k == "constant": return g.const(e.a)
    if k == "fixed": return g.col(qc.fixed + e.a, e.b)
    if k == "advice": return g.col(qc.advice + e.a, e.b)
    if k == "instance": return g.col(qc.instance + e.a, e.b)
    if k == "challenge": return g.const(challenges[e.a])
    if k == "neg": return g.neg(_add_expression(g, cs, qc, e.a, challenges))
    if k == "sum":
        if e.b.kind == "neg":      # the reference folds a + (-b) into Sub
            return g.sub(_add_expression(g, cs, qc, e.a, challenges), _add_expression(g, cs, qc, e.b.a, challenges))
        return g.add(_add_expression(g, cs, qc, e.a, challenges), _add_expression(g, cs, qc, e.b, challenges))
    if k == "product": return g.mul(_add_expression(g, cs, qc, e.a, challenges), _add_expression(g, cs, qc, e.b, challenges))
    if k == "scaled": return g.mul(_add_expression(g, cs, qc, e.a, challenges), g.const(e.b))
    raise ValueError(k)


DELTA = pow(7, 1 << F.S, F.R_MOD)     # `Fr::DELTA` = g^(2^S), generator of the odd-order subgroup cosets


def _evaluate_h_terms(cs: ConstraintSystem, beta: int, gamma: int, theta: int, challenges: Sequence[int], zeta: int):
    """The terms `evaluate_h` folds with y, in the order of [DEP] plonk/evaluation.rs `Evaluator::evaluate_h` (custom gates, permutation,
    lookups), each as a builder `term(g) -> node` so that one graph (the whole quotient numerator) or several (its parts) can be made."""
    qc = quotient_columns(cs)
    last_rotation = -(cs.blinding_factors + 1)
    terms = []
    for polys in cs.gates:                                              # custom gates: Horner(PreviousValue = 0, all gate polynomials in order, y)
        for poly in polys:
            terms.append(lambda g, poly=poly: _add_expression(g, cs, qc, poly, challenges))
    l0 = lambda g: g.col(qc.l0)
    l_last = lambda g: g.col(qc.l_last)
    l_active = lambda g: g.col(qc.l_active_row)
    one = lambda g: g.const(1)
    if cs.permutation_columns:                                          # permutation argument
        sets = cs.num_permutation_sets
        z = lambda g, s_, rot=0: g.col(qc.perm_product + s_, rot)
        terms.append(lambda g: g.mul(g.sub(one(g), z(g, 0)), l0(g)))                                          # l_0 (1 - z_0)
        terms.append(lambda g: g.mul(g.sub(g.mul(z(g, sets - 1), z(g, sets - 1)), z(g, sets - 1)), l_last(g)))  # l_last (z_l^2 - z_l)
        for s_ in range(1, sets):
            terms.append(lambda g, s_=s_: g.mul(g.sub(z(g, s_), z(g, s_ - 1, last_rotation)), l0(g)))          # l_0 (z_i - z_{i-1}(w^last X))
        col_of = {"fixed": qc.fixed, "advice": qc.advice, "instance": qc.instance}

        def perm_set(g, s_):
            # current_delta = beta * zeta * delta^j * extended_omega^row
            x_term = g.mul(g.rowpow(), g.const(beta * zeta % F.R_MOD))
            chunk = cs.permutation_columns[s_ * cs.chunk_len:(s_ + 1) * cs.chunk_len]
            left = z(g, s_, 1)
            for j, (kind, idx) in enumerate(chunk):
                v = g.col(col_of[kind] + idx)
                sig = g.col(qc.sigma + s_ * cs.chunk_len + j)
                left = g.mul(left, g.add(g.mad(sig, g.const(beta), v), g.const(gamma)))          # v + beta sigma + gamma
            right = z(g, s_)
            delta_pow = pow(DELTA, s_ * cs.chunk_len, F.R_MOD)
            for (kind, idx) in chunk:
                v = g.col(col_of[kind] + idx)
                cur = x_term if delta_pow == 1 else g.mul(x_term, g.const(delta_pow))
                right = g.mul(right, g.add(g.add(v, cur), g.const(gamma)))                       # v + delta^j beta X + gamma
                delta_pow = delta_pow * DELTA % F.R_MOD
            return g.mul(g.sub(left, right), l_active(g))

        for s_ in range(sets):
            terms.append(lambda g, s_=s_: perm_set(g, s_))
    for li, lk in enumerate(cs.lookups):                                # lookup arguments
        prod = lambda g, rot=0, li=li: g.col(qc.lookup + 3 * li, rot)
        pin = lambda g, rot=0, li=li: g.col(qc.lookup + 3 * li + 1, rot)
        ptab = lambda g, li=li: g.col(qc.lookup + 3 * li + 2)

        def table_value(g, lk=lk):
            zero, th = g.const(0), g.const(theta)
            cin = g.horner(zero, [_add_expression(g, cs, qc, e, challenges) for e in lk.input_expressions], th)
            ctab = g.horner(zero, [_add_expression(g, cs, qc, e, challenges) for e in lk.table_expressions], th)
            return g.mul(g.add(cin, g.const(beta)), g.add(ctab, g.const(gamma)))

        a_minus_s = lambda g, pin=pin, ptab=ptab: g.sub(pin(g), ptab(g))
        terms.append(lambda g, prod=prod: g.mul(g.sub(one(g), prod(g)), l0(g)))
        terms.append(lambda g, prod=prod: g.mul(g.sub(g.mul(prod(g), prod(g)), prod(g)), l_last(g)))
        terms.append(lambda g, prod=prod, pin=pin, ptab=ptab, table_value=table_value: g.mul(
            g.sub(g.mul(g.mul(prod(g, 1), g.add(pin(g), g.const(beta))), g.add(ptab(g), g.const(gamma))), g.mul(prod(g), table_value(g))), l_active(g)))
        terms.append(lambda g, a_minus_s=a_minus_s: g.mul(a_minus_s(g), l0(g)))
        terms.append(lambda g, a_minus_s=a_minus_s, pin=pin: g.mul(g.mul(a_minus_s(g), g.sub(pin(g), pin(g, -1))), l_active(g)))
    return terms


def _fold_terms(terms, y: int, k: int, extended_k: int) -> RowProgram:
    g = Graph()
    Y = g.const(y)
    value = g.const(0)
    for term in terms:
        value = g.mad(value, Y, term(g))
    return compile_graph(g, value, rot_scale=1 << (extended_k - k), omega=F.omega_for(extended_k))


def evaluate_h_program(cs: ConstraintSystem, k: int, extended_k: int, beta: int, gamma: int, theta: int, y: int,
                       challenges: Sequence[int] = (), zeta: int = F.ZETA) -> RowProgram:
    """The program whose output column is `evaluate_h`'s `values` (before `divide_by_vanishing_poly`).
    Terms and their order follow [DEP] plonk/evaluation.rs `Evaluator::evaluate_h`: custom gates, permutation, lookups."""
    return _fold_terms(_evaluate_h_terms(cs, beta, gamma, theta, challenges, zeta), y, k, extended_k)


def evaluate_h_parts(cs: ConstraintSystem, k: int, extended_k: int, beta: int, gamma: int, theta: int, y: int, parts: int,
                     challenges: Sequence[int] = (), zeta: int = F.ZETA) -> Tuple[List[RowProgram], List[int]]:
    """The same quotient numerator as a SUM of `parts` programs over consecutive runs of its terms: values = sum_p weights[p] * program_p
    with weights[p] = y^(number of terms after part p) -- `zkhip_fr_eval_rows_sum_device` runs the programs side by side.  A circuit with
    hundreds of columns at 2^13 .. 2^15 rows has thousands of instructions and few rows: one program is a handful of wavefronts walking
    the whole list, `parts` programs fill the chip.  (The y-fold is linear in the terms: h = sum_i term_i y^(T-1-i).)"""
    terms = _evaluate_h_terms(cs, beta, gamma, theta, challenges, zeta)
    T = len(terms)
    parts = max(1, min(parts, T))
    # cut where the running instruction count crosses the next multiple of total / parts (a permutation set is ten times a gate polynomial)
    cost = []
    for term in terms:
        g = Graph()
        term(g)
        cost.append(1 + sum(1 for nd in g.nodes if nd[0] == "op"))
    total, run, bounds = sum(cost), 0, [0]
    for i, c in enumerate(cost):
        run += c
        left, need = T - (i + 1), parts - len(bounds)                   # terms after this one, cuts still to make
        if 0 < need and 0 < left and ((run * parts >= total * len(bounds) and left >= need) or left == need):
            bounds.append(i + 1)
    bounds.append(T)
    parts = len(bounds) - 1
    progs, weights = [], []
    for p in range(parts):
        lo, hi = bounds[p], bounds[p + 1]
        progs.append(_fold_terms(terms[lo:hi], y, k, extended_k))
        weights.append(pow(y, T - hi, F.R_MOD))
    return progs, weights


def run_programs_sum_device(progs: Sequence[RowProgram], weights: Sequence[int], columns: Sequence[int], log_rows: int, out: int, stream: int = 0) -> None:
    """out[row] = sum_p weights[p] * progs[p](row) over device-resident columns (zkhip_fr_eval_rows_sum_device)"""
    lib = _lib.load()
    marshalled = [p._marshal() for p in progs]
    arr = (_lib.VmProgram * len(progs))(*[m[0] for m in marshalled])
    w = F.fr_encode(list(weights))
    ptrs = (C.c_void_p * max(len(columns), 1))(*columns)
    _lib.check(lib.zkhip_fr_eval_rows_sum_device(arr, w.ctypes.data, len(progs), ptrs, len(columns), log_rows, C.c_void_p(out), C.c_void_p(stream)))


def linear_combination_program(coeffs: Sequence[int]) -> RowProgram:
    """out[row] = sum_k coeffs[k] * column_k[row]  (multiopen: the per-point polynomial combinations)."""
    g = Graph()
    acc = g.mul(g.col(0), g.const(coeffs[0]))
    for kk in range(1, len(coeffs)):
        acc = g.mad(g.col(kk), g.const(coeffs[kk]), acc)
    return compile_graph(g, acc)


# ---------------------------------------------------------------------------------------------------
# grand-product inputs ([DEP] plonk/permutation/prover.rs `Argument::commit`, plonk/lookup/prover.rs `commit_product`):
# numerator / denominator columns over the base domain (Lagrange basis, 2^k rows), fed to zkhip_fr_grand_product
# ---------------------------------------------------------------------------------------------------
def permutation_denominator_program(n_cols: int, beta: int, gamma: int) -> RowProgram:
    """columns [0, n) = the chunk's values, [n, 2n) = its permutation columns sigma_j:  prod_j (v_j + beta sigma_j + gamma)"""
    g = Graph()
    B, G = g.const(beta), g.const(gamma)
    acc = None
    for j in range(n_cols):
        t = g.add(g.mad(g.col(n_cols + j), B, g.col(j)), G)
        acc = t if acc is None else g.mul(acc, t)
    return compile_graph(g, acc)


def permutation_numerator_program(n_cols: int, first_col: int, beta: int, gamma: int, k: int) -> RowProgram:
    """columns [0, n) = the chunk's values; the chunk starts at permutation column `first_col`:
    prod_j (v_j + delta^(first_col + j) beta omega^row + gamma)"""
    g = Graph()
    G = g.const(gamma)
    acc = None
    for j in range(n_cols):
        t = g.add(g.mad(g.rowpow(), g.const(pow(DELTA, first_col + j, F.R_MOD) * beta % F.R_MOD), g.col(j)), G)
        acc = t if acc is None else g.mul(acc, t)
    return compile_graph(g, acc, omega=F.omega_for(k))


def lookup_product_programs(n_inputs: int, n_tables: int, beta: int, gamma: int, theta: int) -> Tuple[RowProgram, RowProgram]:
    """numerator over columns [inputs..., tables...]: (theta-compressed inputs + beta)(theta-compressed tables + gamma);
    denominator over columns [permuted_input, permuted_table]: (a' + beta)(s' + gamma)"""
    g = Graph()
    T = g.const(theta)
    cin = g.horner(g.const(0), [g.col(i) for i in range(n_inputs)], T)
    ctab = g.horner(g.const(0), [g.col(n_inputs + i) for i in range(n_tables)], T)
    num = compile_graph(g, g.mul(g.add(cin, g.const(beta)), g.add(ctab, g.const(gamma))))
    g2 = Graph()
    den = compile_graph(g2, g2.mul(g2.add(g2.col(0), g2.const(beta)), g2.add(g2.col(1), g2.const(gamma))))
    return num, den


def permute_expression_pair(input_expression: np.ndarray, table_expression: np.ndarray, usable_rows: int):
    """`lookup::prover::permute_expression_pair` [DEP plonk/lookup/prover.rs] on the usable rows: ((n,4) uint64, (n,4) uint64) ->
    (permuted_input, permuted_table), each (usable_rows, 4) uint64.  Raises ZkhipError when an input value is not in the table
    (the reference's Error::ConstraintSystemFailure).  The caller appends the blinding rows."""
    a = np.ascontiguousarray(input_expression, dtype=np.uint64).reshape(-1, 4)
    s = np.ascontiguousarray(table_expression, dtype=np.uint64).reshape(-1, 4)
    assert a.shape[0] >= usable_rows and s.shape[0] >= usable_rows
    pa = np.zeros((usable_rows, 4), dtype=np.uint64)
    ps = np.zeros((usable_rows, 4), dtype=np.uint64)
    _lib.check(_lib.load().zkhip_lookup_permute(a.ctypes.data, s.ctypes.data, usable_rows, pa.ctypes.data, ps.ctypes.data))
    return pa, ps


# ---------------------------------------------------------------------------------------------------
# the row programs of one proof, written out for a host that is not Python (rust-shim/prover_patch.rs, tests/cpp/prover_sequence.c)
# ---------------------------------------------------------------------------------------------------
def halo2_lib_shape(gate_cols: int, lookups: int, blinding: int = 5) -> ConstraintSystem:
    """The halo2-lib `BaseConfig` shape the reference's circuits use (/root/reference/aggregator/benches/wrapper_circuit.rs:61-68,
    state_transition_circuit.rs:48-50): `gate_cols` advice columns with the vertical gate q (a + b c - d), `lookups` range-lookup advice
    columns against one table, one constants column; every advice column and the constants column take part in the permutation."""
    G, NL = gate_cols, lookups
    return ConstraintSystem(
        num_fixed=G + 2, num_advice=G + NL, num_instance=0,
        gates=[[Fixed(i) * (Advice(i, 0) + Advice(i, 1) * Advice(i, 2) - Advice(i, 3))] for i in range(G)],
        lookups=[Lookup([Advice(G + j)], [Fixed(G + 1)]) for j in range(NL)],
        permutation_columns=[("advice", i) for i in range(G + NL)] + [("fixed", G)], blinding_factors=blinding, degree=4)


def export_prover_programs(k: int, gate_cols: int, lookups: int, seed: int = 1) -> bytes:
    """Everything tests/cpp/prover_sequence.c needs for one proof of the halo2-lib shape at 2^k rows: the domain constants and the row
    programs (challenges seeded -- there is no transcript on this side), in the order the C program reads them."""
    import random
    import struct

    from .domain import EvaluationDomain

    cs = halo2_lib_shape(gate_cols, lookups)
    qc = quotient_columns(cs)
    dom = EvaluationDomain(4, k)
    rng = random.Random(seed)
    beta, gamma, theta, y, x = (rng.randrange(1, F.R_MOD) for _ in range(5))
    w = lambda a: np.ascontiguousarray(a).astype("<u8").tobytes()
    wk, usable = F.omega_for(k), (1 << k) - (cs.blinding_factors + 1)
    rot_points = [x * pow(wk, r, F.R_MOD) % F.R_MOD for r in (0, 1, 2, 3, -1, usable)]      # the opening points of the multi-open argument
    y_mo, v_mo, u_mo = (rng.randrange(1, F.R_MOD) for _ in range(3))
    out = [struct.pack("<4sI", b"ZKPS", 3),
           struct.pack("<8I", k, dom.extended_k, gate_cols, lookups, len(cs.permutation_columns), cs.num_permutation_sets, cs.chunk_len, cs.blinding_factors),
           struct.pack("<7I", qc.total, qc.fixed, qc.advice, qc.l0, qc.sigma, qc.perm_product, qc.lookup),
           w(dom.omega_inv), w(dom.ifft_divisor), w(dom.extended_omega), w(dom.extended_omega_inv), w(dom.extended_ifft_divisor), w(dom.g_coset),
           w(F.fr_encode([x])[0]), w(F.fr_encode([F.omega_for(k)])[0]), w(F.fr_encode([DELTA])[0]), w(F.fr_encode([beta])[0]), w(F.fr_encode([gamma])[0]),
           w(F.fr_encode(rot_points)), w(F.fr_encode([y_mo, v_mo, u_mo])),
           struct.pack("<I", dom.t_evaluations.shape[0]), w(dom.t_evaluations)]      # (version 2: omega, delta, beta, gamma; version 3: x omega^r for r = 0, 1, 2, 3, -1, u and the multi-open challenges)
    to_mont = RowProgram()                # raw integer words are the Montgomery form of a / R: multiply by R
    to_mont.emit(OP_MUL, 0, to_mont.column(0), to_mont.constant(pow(2, 256, F.R_MOD)))
    progs = [to_mont]
    for si in range(cs.num_permutation_sets):
        lo, hi = si * cs.chunk_len, min((si + 1) * cs.chunk_len, len(cs.permutation_columns))
        progs.append(permutation_numerator_program(hi - lo, lo, beta, gamma, k))
    for si in range(cs.num_permutation_sets):
        lo, hi = si * cs.chunk_len, min((si + 1) * cs.chunk_len, len(cs.permutation_columns))
        progs.append(permutation_denominator_program(hi - lo, beta, gamma))
    progs += list(lookup_product_programs(1, 1, beta, gamma, theta))
    progs.append(evaluate_h_program(cs, k, dom.extended_k, beta, gamma, theta, y))
    # the same quotient numerator as a sum of programs (zkhip_fr_eval_rows_sum_device), used by the device-resident sequence when the program is
    # long and the rows are few: u32 count (0 = not worth it), count weights, count programs
    tail = [struct.pack("<I", 0)]
    if len(progs[-1].insns) > 512 and dom.extended_k < 18:
        parts, weights = evaluate_h_parts(cs, k, dom.extended_k, beta, gamma, theta, y, 16)
        tail = [struct.pack("<I", len(parts)), w(F.fr_encode(weights))] + [p.to_bytes() for p in parts]
    return b"".join(out + [p.to_bytes() for p in progs] + tail)
