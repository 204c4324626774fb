"""`keygen_vk` / `keygen_pk` and the `VerifyingKey` / `ProvingKey` file formats.

Mirror of [DEP] halo2-axiom `plonk/keygen.rs`, `plonk/permutation/keygen.rs` and the `read` / `write` methods of `plonk.rs` /
`plonk/permutation.rs`, as the reference reaches them:
  * `keygen_vk` + `keygen_pk`                     /root/reference/aggregator/src/wrapper.rs:106-109, :236-242 (and inside the timed loop
                                                   of the wrapper bench, /root/reference/aggregator/benches/wrapper_circuit.rs:123-141)
  * `pk.write(&mut f, SerdeFormat::RawBytesUnchecked)`  /root/reference/aggregator/src/wrapper.rs:967-989 (`build/*_pk.bin`)
  * `ProvingKey::read::<_, C>(.., RawBytesUnchecked, ..)` /root/reference/aggregator/src/wrapper.rs:1007-1034, :1073-1106

The dependency is un-vendored and no key file exists in the reference tree, so the layout is the PUBLISHED one restated from the
crate -- **unpinned** (DESIGN.md section 6): a key written by the Rust prover has never been read by this code.

    VerifyingKey:  k (u32 BE) | #fixed_commitments (u32 BE) | fixed_commitments (G1Affine each) |
                   permutation commitments (one G1Affine per permutation column, no count) |
                   selectors: for every selector ceil(n / 8) bytes, 8 rows per byte, row j of a chunk in bit j
    Polynomial:    #values (u32 BE) | values (Fr each)
    ProvingKey:    VerifyingKey | l0 | l_last | l_active_row (extended-coset polynomials) |
                   fixed_values | fixed_polys | fixed_cosets (each: count u32 BE, then polynomials) |
                   permutation: permutations | polys | cosets (each: count u32 BE, then polynomials)
    Fr / G1Affine under RawBytes / RawBytesUnchecked: the in-memory Montgomery limbs (4 x u64 LE per field element; a point is x || y),
    i.e. exactly the arrays the C ABI takes.  RawBytes checks on read that every element is canonical and every point on the curve;
    RawBytesUnchecked does not.  `Processed` stores commitments compressed (32 B, srs.py) and scalars as their canonical integers
    (`to_repr`, 32 B little-endian): both conversions run on the GPU.

What keygen computes -- everything on the GPU through the batched device entry points:
    fixed commitments / permutation commitments   `params.commit_lagrange(column)`                      prepared-table MSM
    fixed_polys, permutation polys, l0 ...        `domain.lagrange_to_coeff`     zkhip_ifft_scaled_batch_device
    fixed_cosets, permutation cosets, l0 ...      `domain.coeff_to_extended`     zkhip_coeff_to_extended_device
    sigma columns                                 delta^col' * omega^row' from the copy-constraint cycles (`Assembly`), one row program
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass, field
from typing import BinaryIO, List, Optional, Sequence

import numpy as np

from . import _lib, evaluation as E
from .domain import EvaluationDomain
from .fields import Q_MOD, R_MOD, fr_encode

RAW_BYTES, RAW_BYTES_UNCHECKED, PROCESSED = "RawBytes", "RawBytesUnchecked", "Processed"      # `SerdeFormat`


# ---------------------------------------------------------------------------------------------------
# permutation::keygen::Assembly
# ---------------------------------------------------------------------------------------------------
class Assembly:
    """Copy-constraint cycles over the permutation columns: `mapping[col][row]` is the next cell of the cycle through (col, row).
    `copy` merges two cycles exactly as the reference does (smaller into larger, then one swap of the two mapping entries), so the
    resulting sigma polynomials are the reference's for the same sequence of `copy` calls."""

    def __init__(self, n: int, n_columns: int):
        self.n, self.n_columns = n, n_columns
        self.map_col = np.repeat(np.arange(n_columns, dtype=np.int32)[:, None], n, axis=1)
        self.map_row = np.repeat(np.arange(n, dtype=np.int32)[None, :], n_columns, axis=0)
        self.aux_col, self.aux_row = self.map_col.copy(), self.map_row.copy()
        self.sizes = np.ones((n_columns, n), dtype=np.int64)

    def copy(self, left_column: int, left_row: int, right_column: int, right_row: int) -> None:
        if not (0 <= left_column < self.n_columns and 0 <= right_column < self.n_columns):
            raise ValueError("column is not part of the permutation argument")       # Error::ColumnNotInPermutation
        if not (0 <= left_row < self.n and 0 <= right_row < self.n):
            raise ValueError("row out of bounds")                                      # Error::BoundsFailure
        lc, lr = int(self.aux_col[left_column, left_row]), int(self.aux_row[left_column, left_row])
        rc, rr = int(self.aux_col[right_column, right_row]), int(self.aux_row[right_column, right_row])
        if (lc, lr) == (rc, rr):
            return                                                                     # already in one cycle
        if self.sizes[lc, lr] < self.sizes[rc, rr]:
            (lc, lr), (rc, rr) = (rc, rr), (lc, lr)
        self.sizes[lc, lr] += self.sizes[rc, rr]
        i = (rc, rr)
        while True:                                                                    # the right cycle takes the left cycle's representative
            self.aux_col[i], self.aux_row[i] = lc, lr
            i = (int(self.map_col[i]), int(self.map_row[i]))
            if i == (rc, rr):
                break
        a, b = (left_column, left_row), (right_column, right_row)
        ac, ar, bc, br = self.map_col[a], self.map_row[a], self.map_col[b], self.map_row[b]
        self.map_col[a], self.map_row[a], self.map_col[b], self.map_row[b] = bc, br, ac, ar

    def sigma_columns(self, k: int) -> List[np.ndarray]:
        """`build_pk`'s `permutations`: sigma_i[j] = delta^(mapping col) * omega^(mapping row), Lagrange basis, one (n, 4) array per
        column.  omega^row comes from one row program; the products run as one more (two gathered columns multiplied)."""
        n = 1 << k
        assert n == self.n
        omega = _omega(k)
        powers = E.RowProgram(omega=omega)
        powers.emit(E.OP_MOV, 0, E.RowProgram.ROWPOW)
        omega_pow = powers.run([], k)                                                  # omega^j
        deltas = fr_encode([pow(E.DELTA, c, R_MOD) for c in range(self.n_columns)])
        mul = E.RowProgram()
        mul.emit(E.OP_MUL, 0, mul.column(0), mul.column(1))
        out = []
        for c in range(self.n_columns):
            out.append(mul.run([omega_pow[self.map_row[c]], deltas[self.map_col[c]]], k))
        return out


def _omega(k: int) -> int:
    from .fields import omega_for

    return omega_for(k)


# ---------------------------------------------------------------------------------------------------
# device helpers (no torch in the package: buffers come from the C ABI)
# ---------------------------------------------------------------------------------------------------
class _DeviceBuffer:
    def __init__(self, nbytes: int):
        self.lib = _lib.load()
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        _lib.check(self.lib.zkhip_alloc(nbytes, C.byref(self.ptr)))

    def upload(self, a: np.ndarray, offset: int = 0) -> None:
        a = np.ascontiguousarray(a)
        _lib.check(self.lib.zkhip_upload(C.c_void_p(self.ptr.value + offset), a.ctypes.data, a.nbytes))

    def download(self, shape, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=np.uint64)
        _lib.check(self.lib.zkhip_download(out.ctypes.data, C.c_void_p(self.ptr.value + offset), out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            self.lib.zkhip_free(self.ptr)
            self.ptr = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()


def _transform_columns(dom: EvaluationDomain, lagrange: Sequence[np.ndarray], want_cosets: bool = True, max_bytes: int = 1 << 32):
    """(polys, cosets) of Lagrange-basis columns: batched `lagrange_to_coeff` then batched `coeff_to_extended`, in groups of columns
    that keep the device buffers under `max_bytes`."""
    lib = _lib.load()
    n, en = dom.n, dom.extended_len()
    polys, cosets = [], []
    per_col = n * 32 + (en * 32 if want_cosets else 0)
    group = max(1, min(len(lagrange), max_bytes // per_col)) if lagrange else 1
    for g0 in range(0, len(lagrange), group):
        cols = lagrange[g0:g0 + group]
        m = len(cols)
        with _DeviceBuffer(m * n * 32) as d_c:
            d_c.upload(np.stack([np.ascontiguousarray(c, dtype=np.uint64).reshape(n, 4) for c in cols]))
            _lib.check(lib.zkhip_ifft_scaled_batch_device(d_c.ptr, dom.omega_inv.ctypes.data, dom.k, dom.ifft_divisor.ctypes.data, m, n, None))
            pc = d_c.download((m, n, 4))
            polys.extend(pc[i] for i in range(m))
            if want_cosets:
                with _DeviceBuffer(m * en * 32) as d_e:
                    _lib.check(lib.zkhip_coeff_to_extended_device(d_c.ptr, n, dom.k, d_e.ptr, en, dom.extended_k, m, dom.extended_omega.ctypes.data,
                                                                  dom.g_coset.ctypes.data, None))
                    ec = d_e.download((m, en, 4))
                    cosets.extend(ec[i] for i in range(m))
    return polys, cosets


# ---------------------------------------------------------------------------------------------------
# keys
# ---------------------------------------------------------------------------------------------------
@dataclass
class VerifyingKey:
    k: int
    fixed_commitments: np.ndarray                       # (num_fixed, 8) uint64, G1Affine
    permutation_commitments: np.ndarray                 # (num permutation columns, 8) uint64
    selectors: List[np.ndarray] = field(default_factory=list)      # bool arrays of n rows
    cs: Optional[E.ConstraintSystem] = None             # re-derived from the circuit on read (not serialised), as in the reference

    def write(self, w: BinaryIO, fmt: str = RAW_BYTES) -> None:
        _check_format(fmt)
        w.write(struct.pack(">I", self.k))
        w.write(struct.pack(">I", self.fixed_commitments.shape[0]))
        _write_points(w, self.fixed_commitments, fmt)
        _write_points(w, self.permutation_commitments, fmt)
        for sel in self.selectors:
            w.write(np.packbits(np.asarray(sel, dtype=bool), bitorder="little").tobytes())

    @classmethod
    def read(cls, r: BinaryIO, fmt: str, cs: E.ConstraintSystem, num_selectors: int = 0) -> "VerifyingKey":
        """`VerifyingKey::read::<_, ConcreteCircuit>`: the constraint system (column counts, selectors) comes from the circuit, the
        commitments from the file"""
        _check_format(fmt)
        k, = struct.unpack(">I", _take(r, 4))
        if k > 28:
            raise ValueError(f"circuit size k = {k} out of range")
        nf, = struct.unpack(">I", _take(r, 4))
        if nf != cs.num_fixed:
            raise ValueError(f"the key holds {nf} fixed commitments, the circuit has {cs.num_fixed} fixed columns")
        fixed = _read_points(r, nf, fmt)
        perm = _read_points(r, len(cs.permutation_columns), fmt)
        n = 1 << k
        sels = [np.unpackbits(np.frombuffer(_take(r, (n + 7) // 8), dtype=np.uint8), bitorder="little")[:n].astype(bool) for _ in range(num_selectors)]
        return cls(k, fixed, perm, sels, cs)


@dataclass
class ProvingKey:
    vk: VerifyingKey
    l0: np.ndarray                                      # extended coset (2^extended_k, 4)
    l_last: np.ndarray
    l_active_row: np.ndarray
    fixed_values: List[np.ndarray]                      # Lagrange basis (n, 4)
    fixed_polys: List[np.ndarray]                       # coefficients (n, 4)
    fixed_cosets: List[np.ndarray]                      # extended coset
    permutations: List[np.ndarray]                      # sigma columns, Lagrange basis
    permutation_polys: List[np.ndarray]
    permutation_cosets: List[np.ndarray]

    def write(self, w: BinaryIO, fmt: str = RAW_BYTES_UNCHECKED) -> None:
        _check_format(fmt)
        self.vk.write(w, fmt)
        for p in (self.l0, self.l_last, self.l_active_row):
            _write_poly(w, p, fmt)
        for group in (self.fixed_values, self.fixed_polys, self.fixed_cosets, self.permutations, self.permutation_polys, self.permutation_cosets):
            w.write(struct.pack(">I", len(group)))
            for p in group:
                _write_poly(w, p, fmt)

    @classmethod
    def read(cls, r: BinaryIO, fmt: str, cs: E.ConstraintSystem, num_selectors: int = 0) -> "ProvingKey":
        vk = VerifyingKey.read(r, fmt, cs, num_selectors)
        n = 1 << vk.k
        dom = EvaluationDomain(cs.degree, vk.k)
        en = dom.extended_len()
        l0, l_last, l_active = (_read_poly(r, fmt, en) for _ in range(3))
        groups = []
        for want_len, want_count in ((n, cs.num_fixed), (n, cs.num_fixed), (en, cs.num_fixed), (n, len(cs.permutation_columns)),
                                     (n, len(cs.permutation_columns)), (en, len(cs.permutation_columns))):
            count, = struct.unpack(">I", _take(r, 4))
            if count != want_count:
                raise ValueError(f"the key holds {count} polynomials where the circuit has {want_count} columns")
            groups.append([_read_poly(r, fmt, want_len) for _ in range(count)])
        return cls(vk, l0, l_last, l_active, *groups)

    # the column block `evaluate_h_program` reads from the proving key, in QuotientColumns order
    def quotient_key_columns(self):
        return {"fixed": self.fixed_cosets, "l0": self.l0, "l_last": self.l_last, "l_active_row": self.l_active_row, "sigma": self.permutation_cosets}


def keygen_vk(params, cs: E.ConstraintSystem, fixed: Sequence[np.ndarray], assembly: Assembly, selectors: Sequence[np.ndarray] = ()) -> VerifyingKey:
    """`keygen_vk(params, circuit)`: commitments to the fixed columns and to the sigma columns (Lagrange-basis MSMs against the
    registered `g_lagrange`).  `fixed` already holds the selector-derived columns (the mirror has no floor planner)."""
    n = 1 << params.k
    if len(fixed) != cs.num_fixed or assembly.n_columns != len(cs.permutation_columns) or assembly.n != n:
        raise ValueError("fixed columns / permutation assembly do not match the constraint system")
    if n < cs.blinding_factors + 3:
        raise ValueError("not enough rows available")                                  # Error::not_enough_rows_available
    sigma = assembly.sigma_columns(params.k)
    fc = _commit_lagrange_affine(params, fixed)
    pc = _commit_lagrange_affine(params, sigma)
    return VerifyingKey(params.k, fc, pc, [np.asarray(s, dtype=bool) for s in selectors], cs)


def keygen_pk(params, vk: VerifyingKey, cs: E.ConstraintSystem, fixed: Sequence[np.ndarray], assembly: Assembly) -> ProvingKey:
    """`keygen_pk(params, vk, circuit)`: fixed_polys / fixed_cosets, the permutation's polys / cosets, and the l0 / l_last /
    l_active_row cosets.  l_active_row = 1 - (l_last + l_blind) on the extended coset equals the coset of the indicator of the
    usable rows (both are the same polynomial of degree < n), which is how it is produced here."""
    k, n = params.k, 1 << params.k
    if vk.k != k:
        raise ValueError("verifying key and params differ in k")
    dom = EvaluationDomain(cs.degree, k)
    u = n - (cs.blinding_factors + 1)
    one = fr_encode([1])[0]
    l0 = np.zeros((n, 4), dtype=np.uint64); l0[0] = one
    l_last = np.zeros((n, 4), dtype=np.uint64); l_last[u] = one
    l_active = np.zeros((n, 4), dtype=np.uint64); l_active[:u] = one
    sigma = assembly.sigma_columns(k)
    fixed = [np.ascontiguousarray(c, dtype=np.uint64).reshape(n, 4) for c in fixed]
    polys, cosets = _transform_columns(dom, list(fixed) + sigma + [l0, l_last, l_active])
    nf, ns = len(fixed), len(sigma)
    return ProvingKey(vk, cosets[nf + ns], cosets[nf + ns + 1], cosets[nf + ns + 2], fixed, polys[:nf], cosets[:nf],
                      sigma, polys[nf:nf + ns], cosets[nf:nf + ns])


# ---------------------------------------------------------------------------------------------------
# serialisation helpers
# ---------------------------------------------------------------------------------------------------
def _check_format(fmt: str) -> None:
    if fmt not in (RAW_BYTES, RAW_BYTES_UNCHECKED, PROCESSED):
        raise ValueError(f"unknown SerdeFormat {fmt!r}")


# Montgomery words <-> canonical integers as one row program each: a Montgomery product with the constant whose WORDS are 1 (the value
# R^-1) strips the factor R, one with the constant whose words are R^2 (the value R) puts it back
_R_INV = pow(1 << 256, -1, R_MOD)
_R = (1 << 256) % R_MOD


def _convert_repr(p: np.ndarray, constant: int) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4)
    m = p.shape[0]
    if m == 0:
        return p
    log = (m - 1).bit_length()
    if (1 << log) != m:                                    # row programs run over 2^log rows
        pad = np.zeros(((1 << log), 4), dtype=np.uint64)
        pad[:m] = p
        p = pad
    prog = E.RowProgram()
    prog.emit(E.OP_MUL, 0, prog.column(0), prog.constant(constant))
    return prog.run([p], log)[:m]


def _take(r: BinaryIO, nbytes: int) -> bytes:
    b = r.read(nbytes)
    if len(b) != nbytes:
        raise EOFError(f"key file truncated: wanted {nbytes} bytes, got {len(b)}")
    return b


def _write_poly(w: BinaryIO, p: np.ndarray, fmt: str = RAW_BYTES) -> None:
    p = np.ascontiguousarray(p, dtype="<u8").reshape(-1, 4)
    w.write(struct.pack(">I", p.shape[0]))
    w.write((_convert_repr(p, _R_INV) if fmt == PROCESSED else p).tobytes())


def _write_points(w: BinaryIO, pts: np.ndarray, fmt: str) -> None:
    if fmt == PROCESSED:
        from .srs import g1_compress

        w.write(g1_compress(pts))
    else:
        w.write(np.ascontiguousarray(pts, dtype="<u8").tobytes())


_R_LIMBS = np.array([(R_MOD >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
_Q_LIMBS = np.array([(Q_MOD >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def _all_below(a: np.ndarray, mod_limbs: np.ndarray) -> bool:
    """every row of the (m, 4) little-endian limb array is < modulus"""
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < mod_limbs[i])
        eq &= a[:, i] == mod_limbs[i]
    return bool(lt.all())


def _read_poly(r: BinaryIO, fmt: str, want_len: int) -> np.ndarray:
    m, = struct.unpack(">I", _take(r, 4))
    if m != want_len:
        raise ValueError(f"polynomial of {m} values where {want_len} were expected")
    a = np.frombuffer(_take(r, m * 32), dtype="<u8").reshape(m, 4).astype(np.uint64)
    if fmt in (RAW_BYTES, PROCESSED) and not _all_below(a, _R_LIMBS):
        raise ValueError("non-canonical field element in key file")
    return _convert_repr(a, _R) if fmt == PROCESSED else a


def _read_points(r: BinaryIO, count: int, fmt: str) -> np.ndarray:
    if fmt == PROCESSED:
        from .srs import g1_decompress

        return g1_decompress(_take(r, count * 32), count, what="key file") if count else np.zeros((0, 8), dtype=np.uint64)
    a = np.frombuffer(_take(r, count * 64), dtype="<u8").reshape(count, 8).astype(np.uint64)
    if fmt == RAW_BYTES and count:
        if not (_all_below(a[:, :4], _Q_LIMBS) and _all_below(a[:, 4:], _Q_LIMBS)):
            raise ValueError("non-canonical coordinate in key file")
        from .srs import g1_first_invalid

        bad = g1_first_invalid(a)
        if bad is not None:
            raise ValueError(f"point {bad} of the key file is not a valid curve point")
    return a


# ---------------------------------------------------------------------------------------------------
# device-resident keygen: the key is produced in HBM and stays there (a k = 22 wrapper key is 10.5 GiB of 288)
# ---------------------------------------------------------------------------------------------------
class DeviceProvingKey:
    """`ProvingKey` whose polynomials live in HBM, as the prover's device-resident pipeline wants them: two
    allocations -- extended cosets [l0 | l_last | l_active_row | fixed... | sigma...] and n-element columns
    [fixed values | sigma values | fixed polys | sigma polys] -- addressed through the accessors.  `to_host()` downloads a `ProvingKey`
    (for `write`), `from_host(pk)` uploads one (after `ProvingKey.read`)."""

    def __init__(self, vk: VerifyingKey, cs: E.ConstraintSystem, ext: _DeviceBuffer, base: _DeviceBuffer):
        self.vk, self.cs, self.k, self.n = vk, cs, vk.k, 1 << vk.k
        self.en = EvaluationDomain(cs.degree, vk.k).extended_len()
        self.nf, self.np = cs.num_fixed, len(cs.permutation_columns)
        self._ext, self._base = ext, base

    # ---- addresses -----------------------------------------------------------------------------------------------------------------
    def _e(self, i: int) -> int:
        return self._ext.ptr.value + i * self.en * 32

    def _b(self, i: int) -> int:
        return self._base.ptr.value + i * self.n * 32

    def l0(self) -> int: return self._e(0)
    def l_last(self) -> int: return self._e(1)
    def l_active_row(self) -> int: return self._e(2)
    def fixed_coset(self, i: int) -> int: return self._e(3 + i)
    def permutation_coset(self, i: int) -> int: return self._e(3 + self.nf + i)
    def fixed_values(self, i: int) -> int: return self._b(i)
    def permutation_values(self, i: int) -> int: return self._b(self.nf + i)
    def fixed_poly(self, i: int) -> int: return self._b(self.nf + self.np + i)
    def permutation_poly(self, i: int) -> int: return self._b(2 * self.nf + self.np + i)

    @staticmethod
    def allocate(vk: VerifyingKey, cs: E.ConstraintSystem) -> "DeviceProvingKey":
        n = 1 << vk.k
        en = EvaluationDomain(cs.degree, vk.k).extended_len()
        cols = cs.num_fixed + len(cs.permutation_columns)
        ext = _DeviceBuffer(max(1, (3 + cols) * en * 32))
        try:
            base = _DeviceBuffer(max(1, 2 * cols * n * 32))
        except Exception:
            ext.free()
            raise
        return DeviceProvingKey(vk, cs, ext, base)

    def to_host(self) -> ProvingKey:
        n, en, nf, npc = self.n, self.en, self.nf, self.np
        e = self._ext.download((3 + nf + npc, en, 4))
        b = self._base.download((2 * (nf + npc), n, 4))
        return ProvingKey(self.vk, e[0], e[1], e[2], [b[i] for i in range(nf)], [b[nf + npc + i] for i in range(nf)], [e[3 + i] for i in range(nf)],
                          [b[nf + i] for i in range(npc)], [b[2 * nf + npc + i] for i in range(npc)], [e[3 + nf + i] for i in range(npc)])

    @staticmethod
    def from_host(pk: ProvingKey, cs: E.ConstraintSystem) -> "DeviceProvingKey":
        d = DeviceProvingKey.allocate(pk.vk, cs)
        try:
            for i, a in enumerate([pk.l0, pk.l_last, pk.l_active_row] + list(pk.fixed_cosets) + list(pk.permutation_cosets)):
                d._ext.upload(np.ascontiguousarray(a, dtype=np.uint64), i * d.en * 32)
            for i, a in enumerate(list(pk.fixed_values) + list(pk.permutations) + list(pk.fixed_polys) + list(pk.permutation_polys)):
                d._base.upload(np.ascontiguousarray(a, dtype=np.uint64), i * d.n * 32)
        except Exception:
            d.free()
            raise
        return d

    def free(self) -> None:
        self._ext.free()
        self._base.free()

    def __enter__(self) -> "DeviceProvingKey":
        return self

    def __exit__(self, *exc) -> None:
        self.free()


def _copy_device(dst: int, src: int, log_rows: int) -> None:
    prog = E.RowProgram()
    prog.emit(E.OP_MOV, 0, prog.column(0))
    prog.run_device([src], log_rows, dst)


def _copy_device_range(dst: int, src: int, elements: int) -> None:
    """`elements` field elements, device to device, as power-of-two pieces (largest first): one launch per set bit instead of one per column"""
    off = 0
    for bit in range(elements.bit_length() - 1, -1, -1):
        if elements >> bit & 1:
            _copy_device(dst + off * 32, src + off * 32, bit)
            off += 1 << bit


def _sigma_to_device(assembly: Assembly, k: int, d_out: int) -> None:
    """the sigma columns (Lagrange basis) written to d_out, column after column: omega^row from a row program, then one gather-multiply
    per column (`zkhip_fr_gather_mul_device`); the mapping travels as 2 x 4 bytes per cell instead of 32"""
    lib = _lib.load()
    n, ncol = 1 << k, assembly.n_columns
    if ncol == 0:
        return
    with _DeviceBuffer(n * 32) as d_pow, _DeviceBuffer(ncol * 32) as d_delta, _DeviceBuffer(2 * ncol * n * 4) as d_idx:
        powers = E.RowProgram(omega=_omega(k))
        powers.emit(E.OP_MOV, 0, E.RowProgram.ROWPOW)
        powers.run_device([], k, d_pow.ptr.value)
        d_delta.upload(fr_encode([pow(E.DELTA, c, R_MOD) for c in range(ncol)]))
        # the whole mapping in two uploads ([column][row] u32 each), then one gather-multiply per column
        d_idx.upload(np.ascontiguousarray(np.stack([np.asarray(assembly.map_row[c], dtype=np.uint32) for c in range(ncol)])))
        d_idx.upload(np.ascontiguousarray(np.stack([np.asarray(assembly.map_col[c], dtype=np.uint32) for c in range(ncol)])), ncol * n * 4)
        for c in range(ncol):
            _lib.check(lib.zkhip_fr_gather_mul_device(d_pow.ptr, n, C.c_void_p(d_idx.ptr.value + c * n * 4), d_delta.ptr, ncol,
                                                      C.c_void_p(d_idx.ptr.value + (ncol + c) * n * 4), n, C.c_void_p(d_out + c * n * 32), None))
        _lib.check(lib.zkhip_sync())                      # the buffers are freed on return


def _commit_lagrange_device(params, d_columns: int, count: int) -> np.ndarray:
    """commit_lagrange of `count` device-resident columns (n elements each, back to back) -> affine points: MSMs against the registered
    g_lagrange (`zkhip_msm_g1_registered_batch_device`), normalised together"""
    lib = _lib.load()
    if count == 0:
        return np.zeros((0, 8), dtype=np.uint64)
    n = 1 << params.k
    with _DeviceBuffer(count * (96 + 64)) as d_out:
        # the columns lie back to back: ONE batched call (a launch set for all of them when they are small, overlapped pairs when they are large)
        _lib.check(lib.zkhip_msm_g1_registered_batch_device(params.g_lagrange.ctypes.data, C.c_void_p(d_columns), n, count, n, d_out.ptr, None))
        _lib.check(lib.zkhip_g1_batch_normalize_device(d_out.ptr, count, C.c_void_p(d_out.ptr.value + count * 96), None))
        return d_out.download((count, 8), count * 96)


def keygen_device(params, cs: E.ConstraintSystem, fixed: Sequence[np.ndarray], assembly: Assembly, selectors: Sequence[np.ndarray] = ()) -> DeviceProvingKey:
    """`keygen_vk` + `keygen_pk` with everything but the fixed columns' upload and the 64-byte commitments staying on the device: sigma
    columns by gather, commitments against the registered g_lagrange, polys / cosets by the batched transforms.  The result equals
    `keygen_pk(params, keygen_vk(...), ...)` element for element (`DeviceProvingKey.to_host()`)."""
    lib = _lib.load()
    k, n = params.k, 1 << params.k
    if len(fixed) != cs.num_fixed or assembly.n_columns != len(cs.permutation_columns) or assembly.n != n:
        raise ValueError("fixed columns / permutation assembly do not match the constraint system")
    if n < cs.blinding_factors + 3:
        raise ValueError("not enough rows available")
    dom = EvaluationDomain(cs.degree, k)
    en = dom.extended_len()
    nf, npc = cs.num_fixed, len(cs.permutation_columns)
    cols = nf + npc
    pk = DeviceProvingKey.allocate(VerifyingKey(k, np.zeros((0, 8), dtype=np.uint64), np.zeros((0, 8), dtype=np.uint64), [np.asarray(s, dtype=bool) for s in selectors], cs), cs)
    try:
        for i, c in enumerate(fixed):
            pk._base.upload(np.ascontiguousarray(c, dtype=np.uint64).reshape(n, 4), i * n * 32)
        _sigma_to_device(assembly, k, pk.permutation_values(0) if npc else 0)
        commits = _commit_lagrange_device(params, pk.fixed_values(0), cols)
        pk.vk.fixed_commitments, pk.vk.permutation_commitments = commits[:nf], commits[nf:]
        # values -> coefficients (copy, then the batched inverse transform in place) -> extended cosets
        _copy_device_range(pk._b(cols), pk._b(0), cols * n)
        if cols:
            _lib.check(lib.zkhip_ifft_scaled_batch_device(C.c_void_p(pk._b(cols)), dom.omega_inv.ctypes.data, k, dom.ifft_divisor.ctypes.data, cols, n, None))
            _lib.check(lib.zkhip_coeff_to_extended_device(C.c_void_p(pk._b(cols)), n, k, C.c_void_p(pk._e(3)), en, dom.extended_k, cols, dom.extended_omega.ctypes.data,
                                                          dom.g_coset.ctypes.data, None))
        # l0, l_last, l_active_row: indicator columns, transformed like the others
        u = n - (cs.blinding_factors + 1)
        one = fr_encode([1])[0]
        ind = np.zeros((3, n, 4), dtype=np.uint64)
        ind[0, 0] = one
        ind[1, u] = one
        ind[2, :u] = one
        with _DeviceBuffer(3 * n * 32) as d_l:
            d_l.upload(ind)
            _lib.check(lib.zkhip_ifft_scaled_batch_device(d_l.ptr, dom.omega_inv.ctypes.data, k, dom.ifft_divisor.ctypes.data, 3, n, None))
            _lib.check(lib.zkhip_coeff_to_extended_device(d_l.ptr, n, k, C.c_void_p(pk._e(0)), en, dom.extended_k, 3, dom.extended_omega.ctypes.data,
                                                          dom.g_coset.ctypes.data, None))
            _lib.check(lib.zkhip_sync())
    except Exception:
        pk.free()
        raise
    return pk


def _commit_lagrange_affine(params, columns: Sequence[np.ndarray]) -> np.ndarray:
    """commit_lagrange of every column -> affine points (the verifying key stores `to_affine()` of the commitments): the commitments
    are normalised together on the GPU (`Curve::batch_normalize`, zkhip_g1_batch_normalize)"""
    if not len(columns):
        return np.zeros((0, 8), dtype=np.uint64)
    jac = np.ascontiguousarray(np.stack([params.commit_lagrange(np.ascontiguousarray(col, dtype=np.uint64)) for col in columns]))
    out = np.zeros((jac.shape[0], 8), dtype=np.uint64)
    _lib.check(_lib.load().zkhip_g1_batch_normalize(jac.ctypes.data, jac.shape[0], out.ctypes.data))
    return out
