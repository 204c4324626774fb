"""CPU: the host logic of keygen.py -- the copy-constraint Assembly and the VerifyingKey / ProvingKey file layout (restated from the
published halo2 crate: unpinned, no key file exists in the reference tree; /root/reference/aggregator/src/wrapper.rs:967-989 writes
them, :1007-1034 reads them)."""
import io
import random
import struct

import numpy as np
import pytest

from zksnap_circuits_halo2_amd import evaluation as E, keygen as KG
from zksnap_circuits_halo2_amd.fields import Q_MOD, R_MOD, fr_encode


def rand_fr(rng, m):
    return fr_encode([rng.randrange(R_MOD) for _ in range(m)])


def rand_points(rng, m):
    a = np.zeros((m, 8), dtype=np.uint64)
    for i in range(m):
        for c in range(2):
            v = rng.randrange(Q_MOD)
            a[i, 4 * c:4 * c + 4] = [(v >> (64 * j)) & ((1 << 64) - 1) for j in range(4)]
    return a


def make_key(rng, k=4, num_fixed=3, perm_cols=2, selectors=1):
    cs = E.ConstraintSystem(num_fixed=num_fixed, num_advice=2, permutation_columns=[("advice", i) for i in range(perm_cols)], degree=4)
    n, en = 1 << k, 1 << (k + 2)
    vk = KG.VerifyingKey(k, rand_points(rng, num_fixed), rand_points(rng, perm_cols), [np.array([rng.random() < 0.5 for _ in range(n)]) for _ in range(selectors)], cs)
    pk = KG.ProvingKey(vk, rand_fr(rng, en), rand_fr(rng, en), rand_fr(rng, en),
                       [rand_fr(rng, n) for _ in range(num_fixed)], [rand_fr(rng, n) for _ in range(num_fixed)], [rand_fr(rng, en) for _ in range(num_fixed)],
                       [rand_fr(rng, n) for _ in range(perm_cols)], [rand_fr(rng, n) for _ in range(perm_cols)], [rand_fr(rng, en) for _ in range(perm_cols)])
    return cs, pk


def key_arrays(pk):
    return ([pk.vk.fixed_commitments, pk.vk.permutation_commitments, pk.l0, pk.l_last, pk.l_active_row] + list(pk.vk.selectors) + pk.fixed_values + pk.fixed_polys +
            pk.fixed_cosets + pk.permutations + pk.permutation_polys + pk.permutation_cosets)


def test_proving_key_round_trip_and_layout():
    rng = random.Random(1)
    cs, pk = make_key(rng)
    buf = io.BytesIO()
    pk.write(buf, KG.RAW_BYTES_UNCHECKED)
    raw = buf.getvalue()
    n, en = 16, 64
    # layout: k, count, fixed commitments, permutation commitments, one selector of n / 8 bytes, then the length-prefixed polynomials
    assert struct.unpack(">I", raw[:4])[0] == 4 and struct.unpack(">I", raw[4:8])[0] == 3
    vk_len = 8 + 3 * 64 + 2 * 64 + n // 8
    assert struct.unpack(">I", raw[vk_len:vk_len + 4])[0] == en                       # l0's value count
    assert raw[8:8 + 64] == pk.vk.fixed_commitments[0].astype("<u8").tobytes()       # raw Montgomery limbs, x || y
    assert len(raw) == vk_len + 3 * (4 + en * 32) + 2 * (4 + 3 * (4 + n * 32)) + (4 + 3 * (4 + en * 32)) + 2 * (4 + 2 * (4 + n * 32)) + (4 + 2 * (4 + en * 32))
    back = KG.ProvingKey.read(io.BytesIO(raw), KG.RAW_BYTES_UNCHECKED, cs, num_selectors=1)
    assert back.vk.k == 4
    for a, b in zip(key_arrays(pk), key_arrays(back)):
        assert np.array_equal(a, b)
    # selector bit order: row j of a chunk of eight in bit j
    sel = pk.vk.selectors[0]
    assert raw[8 + 5 * 64] == sum(int(sel[j]) << j for j in range(8))


def test_key_reader_rejects_bad_files():
    rng = random.Random(2)
    cs, pk = make_key(rng, selectors=0)
    buf = io.BytesIO()
    pk.write(buf, KG.RAW_BYTES_UNCHECKED)
    raw = bytearray(buf.getvalue())
    with pytest.raises(EOFError):
        KG.ProvingKey.read(io.BytesIO(bytes(raw[:-5])), KG.RAW_BYTES_UNCHECKED, cs)
    with pytest.raises(ValueError):
        KG.ProvingKey.read(io.BytesIO(bytes(raw)), "Compressed", cs)                  # not a SerdeFormat (Processed needs the GPU: tests/test_gpu_serde.py)
    other = E.ConstraintSystem(num_fixed=4, num_advice=2, permutation_columns=cs.permutation_columns, degree=4)
    with pytest.raises(ValueError):
        KG.ProvingKey.read(io.BytesIO(bytes(raw)), KG.RAW_BYTES_UNCHECKED, other)     # the circuit has another number of fixed columns
    bad = bytearray(raw)
    bad[0:4] = struct.pack(">I", 40)
    with pytest.raises(ValueError):
        KG.ProvingKey.read(io.BytesIO(bytes(bad)), KG.RAW_BYTES_UNCHECKED, cs)
    # RawBytes (checked) refuses a non-canonical scalar that RawBytesUnchecked lets through
    vk_len = 8 + 3 * 64 + 2 * 64
    bad = bytearray(raw)
    bad[vk_len + 4: vk_len + 4 + 32] = b"\xff" * 32
    KG.ProvingKey.read(io.BytesIO(bytes(bad)), KG.RAW_BYTES_UNCHECKED, cs)
    # (the checked reader validates the commitments on the GPU first; without one it fails there -- either way it must not accept the file)
    with pytest.raises(Exception):
        KG.ProvingKey.read(io.BytesIO(bytes(bad)), KG.RAW_BYTES, cs)


def test_assembly_cycles_are_the_equivalence_classes():
    rng = random.Random(3)
    n, cols = 64, 3
    asm = KG.Assembly(n, cols)
    parent = {(c, r): (c, r) for c in range(cols) for r in range(n)}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for _ in range(120):
        a = (rng.randrange(cols), rng.randrange(n))
        b = (rng.randrange(cols), rng.randrange(n))
        asm.copy(a[0], a[1], b[0], b[1])
        parent[find(a)] = find(b)
    # sigma is a permutation of the cells ...
    image = {(int(asm.map_col[c, r]), int(asm.map_row[c, r])) for c in range(cols) for r in range(n)}
    assert len(image) == cols * n
    # ... whose cycles are exactly the classes of the copy constraints
    classes = {}
    for cell in parent:
        classes.setdefault(find(cell), set()).add(cell)
    for members in classes.values():
        start = next(iter(members))
        seen, cur = set(), start
        while cur not in seen:
            seen.add(cur)
            cur = (int(asm.map_col[cur]), int(asm.map_row[cur]))
        assert seen == members
        assert int(asm.sizes[asm.aux_col[start], asm.aux_row[start]]) == len(members)
    with pytest.raises(ValueError):
        asm.copy(0, 0, cols, 0)
    with pytest.raises(ValueError):
        asm.copy(0, n, 0, 0)


def test_assembly_merge_order_matches_the_published_rule():
    """two-cell example worked by hand from `Assembly::copy`: equal sizes keep (left, right) as given, the right cycle adopts the left
    representative, then the two mapping entries are swapped"""
    asm = KG.Assembly(8, 2)
    asm.copy(0, 1, 1, 2)
    assert (asm.map_col[0, 1], asm.map_row[0, 1]) == (1, 2) and (asm.map_col[1, 2], asm.map_row[1, 2]) == (0, 1)
    assert (asm.aux_col[1, 2], asm.aux_row[1, 2]) == (0, 1) and asm.sizes[0, 1] == 2
    asm.copy(1, 5, 0, 1)                       # left is a singleton, right a 2-cycle: the sides swap, the singleton is merged into the cycle
    assert asm.sizes[0, 1] == 3 and (asm.aux_col[1, 5], asm.aux_row[1, 5]) == (0, 1)
    assert (asm.map_col[1, 5], asm.map_row[1, 5]) == (1, 2) and (asm.map_col[0, 1], asm.map_row[0, 1]) == (1, 5)
    asm.copy(1, 2, 1, 5)                       # same cycle already: nothing changes
    assert asm.sizes[0, 1] == 3
