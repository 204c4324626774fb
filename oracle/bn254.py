"""BN254 big-integer oracle for the MSM / NTT hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`zksnap_circuits_halo2_amd/`, `libzkhip.so`) never does.

PARITY UNPINNED by the reference: `/root/reference` holds no golden vector, KAT or fixture at the
`best_multiexp` / `best_fft` boundary and its arithmetic lives in un-vendored,
un-pinned git dependencies (`halo2-axiom` / `halo2curves-axiom`, reached from
`aggregator/src/wrapper.rs:129` `create_proof`, `aggregator/Cargo.toml:7-21`).  This model restates
the *published* algorithms of those crates [DEP] and is pinned instead by (a) public constants and
known answers (EIP-196 `2*G`, `r*G = O`, `(r-1)*G = -G`), (b) self-consistency identities
(`MSM(a, s^i G) = [sum a_i s^i] G`, `iNTT(NTT(a)) = a`, NTT == O(n^2) DFT).

Memory formats restated:
  * `bn256::Fr` / `bn256::Fq`: four little-endian u64 limbs in Montgomery form (x*2^256 mod p).
  * `G1Affine`: x||y (8 limbs), identity = (0, 0).   `G1`: Jacobian x||y||z (12 limbs), identity z = 0.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------------------------
# Field constants (every derived constant is re-derived in the tests)
# ----------------------------------------------------------------------------------------------
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B97091_43E1F593F0000001  # Fr modulus r
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D_3C208C16D87CFD47  # Fq modulus q
MONT_BITS = 256
MONT_R = 1 << MONT_BITS
MASK64 = (1 << 64) - 1

FR_S = 28                      # 2-adicity of r - 1
FR_GENERATOR = 7               # multiplicative generator used by halo2curves [DEP]
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)   # primitive 2^28-th root
# halo2curves `Fr::ZETA` [DEP]; any primitive cube root works, the library takes it as an argument.
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
CURVE_B = 3
G1_GEN = (1, 2)


def mont_inv64(p: int) -> int:
    """-p^{-1} mod 2^64 (the `INV` constant of a 4x64 Montgomery implementation)."""
    return (-pow(p, -1, 1 << 64)) & MASK64


def to_mont(x: int, p: int) -> int:
    return (x % p) * MONT_R % p


def from_mont(x: int, p: int) -> int:
    return x * pow(MONT_R, -1, p) % p


def limbs4(x: int) -> List[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def from_limbs(l: Sequence[int]) -> int:
    v = 0
    for i, w in enumerate(l):
        v |= int(w) << (64 * i)
    return v


# ----------------------------------------------------------------------------------------------
# G2: y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1)  (halo2curves bn256::{Fq2, G2Affine, G2} [DEP]; the alt_bn128 twist of
# EIP-197).  Fq2 elements are (c0, c1) tuples of canonical ints; affine points ((x0, x1), (y0, y1)), None = identity; Jacobian
# points (X, Y, Z) of Fq2 elements.  Pinned to public values in tests/test_oracle.py: the generator of EIP-197, membership of the
# twist curve, r * G2 = identity.
# ----------------------------------------------------------------------------------------------
F2 = Tuple[int, int]
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def f2_add(a: F2, b: F2) -> F2: return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)
def f2_sub(a: F2, b: F2) -> F2: return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)
def f2_mul(a: F2, b: F2) -> F2: return ((a[0] * b[0] - a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)
def f2_scale(a: F2, k: int) -> F2: return (a[0] * k % Q_MOD, a[1] * k % Q_MOD)


def f2_inv(a: F2) -> F2:
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q_MOD)
    return (a[0] * d % Q_MOD, -a[1] * d % Q_MOD)


G2_B = f2_mul((3, 0), f2_inv((9, 1)))
G2_JAC_ID = ((0, 0), (1, 0), (0, 0))


def g2_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), G2_B)


def g2_neg(P):
    return None if P is None else (P[0], f2_sub((0, 0), P[1]))


def g2_jac_double(P):
    X, Y, Z = P
    if Z == (0, 0) or Y == (0, 0):
        return G2_JAC_ID
    A, B = f2_mul(X, X), f2_mul(Y, Y)
    Cc = f2_mul(B, B)
    t = f2_add(X, B)
    D = f2_scale(f2_sub(f2_sub(f2_mul(t, t), A), Cc), 2)
    E = f2_scale(A, 3)
    X3 = f2_sub(f2_mul(E, E), f2_scale(D, 2))
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), f2_scale(Cc, 8))
    return (X3, Y3, f2_scale(f2_mul(Y, Z), 2))


def g2_jac_add(P, Q):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == (0, 0):
        return Q
    if Z2 == (0, 0):
        return P
    Z1Z1, Z2Z2 = f2_mul(Z1, Z1), f2_mul(Z2, Z2)
    U1, U2 = f2_mul(X1, Z2Z2), f2_mul(X2, Z1Z1)
    S1, S2 = f2_mul(f2_mul(Y1, Z2), Z2Z2), f2_mul(f2_mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        return g2_jac_double(P) if S1 == S2 else G2_JAC_ID
    H, Rr = f2_sub(U2, U1), f2_sub(S2, S1)
    HH = f2_mul(H, H)
    HHH, V = f2_mul(H, HH), f2_mul(U1, HH)
    X3 = f2_sub(f2_sub(f2_mul(Rr, Rr), HHH), f2_scale(V, 2))
    Y3 = f2_sub(f2_mul(Rr, f2_sub(V, X3)), f2_mul(S1, HHH))
    return (X3, Y3, f2_mul(f2_mul(Z1, Z2), H))


def g2_to_jac(P):
    return G2_JAC_ID if P is None else (P[0], P[1], (1, 0))


def g2_to_affine(P):
    X, Y, Z = P
    if Z == (0, 0):
        return None
    zi = f2_inv(Z)
    zi2 = f2_mul(zi, zi)
    return (f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)))


def g2_add(P, Q):
    return g2_to_affine(g2_jac_add(g2_to_jac(P), g2_to_jac(Q)))


def g2_scalar_mul(k: int, P):
    k %= R_MOD
    acc, base = G2_JAC_ID, g2_to_jac(P)
    while k:
        if k & 1:
            acc = g2_jac_add(acc, base)
        base = g2_jac_double(base)
        k >>= 1
    return g2_to_affine(acc)


def g2_msm_naive(scalars: Sequence[int], bases) -> object:
    acc = G2_JAC_ID
    for k, P in zip(scalars, bases):
        if P is not None and k % R_MOD:
            acc = g2_jac_add(acc, g2_to_jac(g2_scalar_mul(k, P)))
    return g2_to_affine(acc)


def g2_affine_to_limbs(P) -> List[int]:
    """G2Affine memory: x.c0 | x.c1 | y.c0 | y.c1, each 4 x u64 Montgomery limbs; identity = all zero"""
    if P is None:
        return [0] * 16
    out: List[int] = []
    for v in (P[0][0], P[0][1], P[1][0], P[1][1]):
        out += limbs4(to_mont(v, Q_MOD))
    return out


def g2_jac_from_limbs(l: Sequence[int]):
    """G2 memory (24 limbs: x, y, z as Fq2) -> affine point or None; asserts canonical limbs"""
    v = []
    for c in range(6):
        m = from_limbs(l[4 * c:4 * c + 4])
        assert m < Q_MOD, "non-canonical limbs"
        v.append(from_mont(m, Q_MOD))
    return g2_to_affine(((v[0], v[1]), (v[2], v[3]), (v[4], v[5])))


# ----------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq.  Affine points are (x, y) tuples of canonical ints; None = identity.
# ----------------------------------------------------------------------------------------------
Affine = Optional[Tuple[int, int]]
Jac = Tuple[int, int, int]
JAC_ID: Jac = (0, 1, 0)


def on_curve(P: Affine) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - CURVE_B) % Q_MOD == 0


def neg(P: Affine) -> Affine:
    return None if P is None else (P[0], (-P[1]) % Q_MOD)


def jac_double(P: Jac) -> Jac:
    X, Y, Z = P
    if Z == 0 or Y == 0:
        return JAC_ID
    A = X * X % Q_MOD
    B = Y * Y % Q_MOD
    C = B * B % Q_MOD
    D = 2 * ((X + B) * (X + B) - A - C) % Q_MOD
    E = 3 * A % Q_MOD
    F = E * E % Q_MOD
    X3 = (F - 2 * D) % Q_MOD
    Y3 = (E * (D - X3) - 8 * C) % Q_MOD
    Z3 = 2 * Y * Z % Q_MOD
    return (X3, Y3, Z3)


def jac_add(P: Jac, Q: Jac) -> Jac:
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % Q_MOD
    Z2Z2 = Z2 * Z2 % Q_MOD
    U1 = X1 * Z2Z2 % Q_MOD
    U2 = X2 * Z1Z1 % Q_MOD
    S1 = Y1 * Z2 * Z2Z2 % Q_MOD
    S2 = Y2 * Z1 * Z1Z1 % Q_MOD
    if U1 == U2:
        return jac_double(P) if S1 == S2 else JAC_ID
    H = (U2 - U1) % Q_MOD
    Rr = (S2 - S1) % Q_MOD
    HH = H * H % Q_MOD
    HHH = H * HH % Q_MOD
    V = U1 * HH % Q_MOD
    X3 = (Rr * Rr - HHH - 2 * V) % Q_MOD
    Y3 = (Rr * (V - X3) - S1 * HHH) % Q_MOD
    Z3 = Z1 * Z2 * H % Q_MOD
    return (X3, Y3, Z3)


def to_jac(P: Affine) -> Jac:
    return JAC_ID if P is None else (P[0], P[1], 1)


def to_affine(P: Jac) -> Affine:
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, Q_MOD)
    zi2 = zi * zi % Q_MOD
    return (X * zi2 % Q_MOD, Y * zi2 * zi % Q_MOD)


def add(P: Affine, Q: Affine) -> Affine:
    return to_affine(jac_add(to_jac(P), to_jac(Q)))


def scalar_mul(k: int, P: Affine) -> Affine:
    k %= R_MOD
    acc = JAC_ID
    base = to_jac(P)
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_double(base)
        k >>= 1
    return to_affine(acc)


def batch_to_affine(points: Sequence[Jac]) -> List[Affine]:
    """Montgomery-trick normalisation (one inversion for the whole list)."""
    prefix, acc = [], 1
    for (_, _, Z) in points:
        prefix.append(acc)
        if Z:
            acc = acc * Z % Q_MOD
    inv = pow(acc, -1, Q_MOD)
    out: List[Affine] = [None] * len(points)
    for i in range(len(points) - 1, -1, -1):
        X, Y, Z = points[i]
        if Z == 0:
            continue
        zi = inv * prefix[i] % Q_MOD
        inv = inv * Z % Q_MOD
        zi2 = zi * zi % Q_MOD
        out[i] = (X * zi2 % Q_MOD, Y * zi2 * zi % Q_MOD)
    return out


def structured_srs(s: int, n: int) -> List[Affine]:
    """g[i] = s^i * G (a KZG monomial SRS with a *known* trapdoor)."""
    jac, cur = [], 1
    # fixed-base comb: G * cur via double-and-add is O(n * 254); fine for the small n used in tests
    for _ in range(n):
        jac.append(to_jac(scalar_mul(cur, G1_GEN)))
        cur = cur * s % R_MOD
    return batch_to_affine(jac)


# ----------------------------------------------------------------------------------------------
# MSM.  `msm_naive` is the mathematical definition; `multiexp_serial` / `best_multiexp` restate
# the published halo2-axiom algorithm [DEP] `halo2_proofs/src/arithmetic.rs` (not under
# /root/reference; entered from aggregator/src/wrapper.rs:129 and
# aggregator/benches/wrapper_circuit.rs:140).
# ----------------------------------------------------------------------------------------------
def msm_naive(scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
    assert len(scalars) == len(bases)
    acc = JAC_ID
    for k, P in zip(scalars, bases):
        acc = jac_add(acc, to_jac(scalar_mul(k, P)))
    return to_affine(acc)


def window_bits(n: int) -> int:
    """`c` of multiexp_serial: 1 (n<4) / 3 (n<32) / ceil(ln n)."""
    if n < 4:
        return 1
    if n < 32:
        return 3
    return math.ceil(math.log(n))


def _get_at(segment: int, c: int, k: int) -> int:
    skip_bits = segment * c
    if skip_bits // 8 >= 32:
        return 0
    return (k >> skip_bits) & ((1 << c) - 1) if skip_bits < 256 else 0


def multiexp_serial(scalars: Sequence[int], bases: Sequence[Affine], acc: Jac = JAC_ID) -> Jac:
    c = window_bits(len(bases))
    segments = 256 // c + 1
    for seg in range(segments - 1, -1, -1):
        for _ in range(c):
            acc = jac_double(acc)
        buckets: List[Jac] = [JAC_ID] * ((1 << c) - 1)
        for k, P in zip(scalars, bases):
            d = _get_at(seg, c, k % R_MOD)
            if d:
                buckets[d - 1] = jac_add(buckets[d - 1], to_jac(P))
        running = JAC_ID
        for b in reversed(buckets):
            running = jac_add(running, b)
            acc = jac_add(acc, running)
    return acc


def best_multiexp(scalars: Sequence[int], bases: Sequence[Affine], threads: int = 1) -> Affine:
    assert len(scalars) == len(bases)
    n = len(scalars)
    if n > threads:
        chunk = n // threads
        parts = [multiexp_serial(scalars[i:i + chunk], bases[i:i + chunk]) for i in range(0, n, chunk)]
        acc = JAC_ID
        for p in parts:
            acc = jac_add(acc, p)
        return to_affine(acc)
    return to_affine(multiexp_serial(scalars, bases))


# ----------------------------------------------------------------------------------------------
# NTT over Fr.  `dft_naive` is the definition a[i] <- sum_j a[j] w^(ij); `best_fft` restates the
# published halo2-axiom radix-2 algorithm [DEP] (bit-reverse, twiddle table w^0..w^(n/2-1), DIT).
# ----------------------------------------------------------------------------------------------
def omega_for(log_n: int) -> int:
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), R_MOD)


def dft_naive(a: Sequence[int], omega: int) -> List[int]:
    n = len(a)
    out = []
    for i in range(n):
        wi = pow(omega, i, R_MOD)
        acc, w = 0, 1
        for j in range(n):
            acc = (acc + a[j] * w) % R_MOD
            w = w * wi % R_MOD
        out.append(acc)
    return out


def bitreverse(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def best_fft(a: Sequence[int], omega: int, log_n: int) -> List[int]:
    n = 1 << log_n
    assert len(a) == n
    a = [x % R_MOD for x in a]
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw, w = [], 1
    for _ in range(n // 2):
        tw.append(w)
        w = w * omega % R_MOD
    chunk, tchunk = 2, n // 2
    for _ in range(log_n):
        half = chunk // 2
        for base in range(0, n, chunk):
            for i in range(half):
                t = a[base + half + i] * tw[i * tchunk] % R_MOD
                u = a[base + i]
                a[base + i] = (u + t) % R_MOD
                a[base + half + i] = (u - t) % R_MOD
        chunk *= 2
        tchunk //= 2
    return a


# ----------------------------------------------------------------------------------------------
# EvaluationDomain restatement [DEP] `halo2_proofs/src/poly/domain.rs`.
# ----------------------------------------------------------------------------------------------
class EvaluationDomain:
    """`EvaluationDomain::new(j, k)`: j = constraint-system degree, n = 2^k, coset generator zeta."""

    def __init__(self, j: int, k: int, zeta: int = FR_ZETA):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_omega = omega_for(ek)
        self.extended_omega_inv = pow(self.extended_omega, -1, R_MOD)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R_MOD)
        self.omega_inv = pow(self.omega, -1, R_MOD)
        self.g_coset = zeta
        self.g_coset_inv = zeta * zeta % R_MOD
        self.ifft_divisor = pow(self.n, -1, R_MOD)
        self.extended_ifft_divisor = pow(1 << ek, -1, R_MOD)
        # t(X) = X^n - 1 on the coset zeta * <extended_omega>: periodic with period 2^(ek-k); stored inverted
        orig = pow(zeta, self.n, R_MOD)
        step = pow(self.extended_omega, self.n, R_MOD)
        t, cur = [], orig
        while True:
            t.append(cur)
            cur = cur * step % R_MOD
            if cur == orig:
                break
        assert len(t) == 1 << (ek - k)
        self.t_evaluations = [pow((v - 1) % R_MOD, -1, R_MOD) for v in t]

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def ifft(self, a, omega_inv, log_n, divisor):
        return [x * divisor % R_MOD for x in best_fft(a, omega_inv, log_n)]

    def lagrange_to_coeff(self, a):
        assert len(a) == self.n
        return self.ifft(a, self.omega_inv, self.k, self.ifft_divisor)

    def coeff_to_lagrange(self, a):
        assert len(a) == self.n
        return best_fft(a, self.omega, self.k)

    def distribute_powers_zeta(self, a, into_coset: bool):
        p = [self.g_coset, self.g_coset_inv] if into_coset else [self.g_coset_inv, self.g_coset]
        return [x if i % 3 == 0 else x * p[i % 3 - 1] % R_MOD for i, x in enumerate(a)]

    def coeff_to_extended(self, a):
        assert len(a) == self.n
        a = self.distribute_powers_zeta(a, True) + [0] * (self.extended_len() - self.n)
        return best_fft(a, self.extended_omega, self.extended_k)

    def extended_to_coeff(self, a):
        assert len(a) == self.extended_len()
        a = self.ifft(a, self.extended_omega_inv, self.extended_k, self.extended_ifft_divisor)
        a = self.distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        assert len(a) == self.extended_len()
        m = len(self.t_evaluations)
        return [x * self.t_evaluations[i % m] % R_MOD for i, x in enumerate(a)]


# ----------------------------------------------------------------------------------------------
# Row a7: eval_polynomial / kate_division [DEP halo2_proofs/src/arithmetic.rs], BatchInvert [DEP ff], and the
# grand-product running product [DEP halo2_proofs/src/plonk/permutation/prover.rs]
# ----------------------------------------------------------------------------------------------
def eval_polynomial(poly: Sequence[int], point: int) -> int:
    acc = 0
    for c in reversed(poly):          # Horner, as the reference's serial `evaluate`
        acc = (acc * point + c) % R_MOD
    return acc


def kate_division(a: Sequence[int], b: int) -> List[int]:
    """Quotient of a(X) by (X - b); the remainder is dropped (the reference negates b and runs the same recurrence)."""
    nb = (-b) % R_MOD
    q = [0] * (len(a) - 1)
    tmp = 0
    for i in range(len(a) - 1, 0, -1):
        lead = (a[i] - tmp) % R_MOD
        q[i - 1] = lead
        tmp = lead * nb % R_MOD
    return q


def batch_invert(a: Sequence[int]) -> List[int]:
    """Elementwise inverse, zeros stay zero (ff::BatchInvert semantics)."""
    return [pow(x, -1, R_MOD) if x % R_MOD else 0 for x in a]


def prefix_product(v: Sequence[int]) -> List[int]:
    """z[0] = 1, z[i+1] = z[i] * v[i]; returns z[0..n)."""
    out, cur = [], 1
    for x in v:
        out.append(cur)
        cur = cur * x % R_MOD
    return out


def grand_product(num: Sequence[int], den: Sequence[int]) -> List[int]:
    """z[0] = 1, z[i+1] = z[i] * num[i] / den[i]  ([DEP] plonk/permutation/prover.rs, plonk/lookup/prover.rs); zero
    denominators count as zero (BatchInvert)."""
    inv = batch_invert(den)
    return prefix_product([a * b % R_MOD for a, b in zip(num, inv)])


def permute_expression_pair(input_expression: Sequence[int], table_expression: Sequence[int], usable_rows: int):
    """[DEP] halo2_proofs/src/plonk/lookup/prover.rs `permute_expression_pair`, statement by statement: sort the input, give the
    first row of every run its own value from the table's multiset (a BTreeMap value -> count; missing value = error), then
    iterate the map in ascending order and pop repeated rows from the end of the list for the leftovers.  Returns the usable
    rows only (the reference appends blinding_factors + 1 random rows)."""
    permuted_input = sorted(x % R_MOD for x in input_expression[:usable_rows])
    leftover = {}
    for v in table_expression[:usable_rows]:
        leftover[v % R_MOD] = leftover.get(v % R_MOD, 0) + 1
    permuted_table = [0] * usable_rows
    repeated_input_rows = []
    for row, v in enumerate(permuted_input):
        if row == 0 or v != permuted_input[row - 1]:
            permuted_table[row] = v
            if leftover.get(v, 0) == 0:
                raise ValueError("ConstraintSystemFailure: lookup input value not in table")
            leftover[v] -= 1
        else:
            repeated_input_rows.append(row)
    for coeff in sorted(leftover):                 # BTreeMap iteration order
        for _ in range(leftover[coeff]):
            permuted_table[repeated_input_rows.pop()] = coeff
    assert not repeated_input_rows
    return permuted_input, permuted_table


def permute_expression_pair_np(input_values, table_values, usable_rows: int):
    """The same statement sequence for values below 2^63 (range-check lookups: inputs below 2^lookup_bits, table = the range), vectorised
    with numpy so that the wrapper's size (2^22 rows, /root/reference/aggregator/benches/wrapper_circuit.rs:21,66) is checked in seconds:
    sorted input; the first row of every run takes its own value out of the table's multiset; the leftovers, ascending, go to the repeated
    rows from the LAST one backwards (BTreeMap iteration + Vec::pop).  tests/test_oracle.py pins it to permute_expression_pair above."""
    import numpy as np

    pin = np.sort(np.asarray(input_values[:usable_rows], dtype=np.int64))
    ts = np.sort(np.asarray(table_values[:usable_rows], dtype=np.int64))
    first = np.ones(usable_rows, dtype=bool)
    first[1:] = pin[1:] != pin[:-1]
    distinct = pin[first]
    at = np.searchsorted(ts, distinct, side="left")              # first instance of every distinct input value in the sorted table
    if np.any(at >= usable_rows) or np.any(ts[np.minimum(at, usable_rows - 1)] != distinct):
        raise ValueError("ConstraintSystemFailure: lookup input value not in table")
    keep = np.ones(usable_rows, dtype=bool)
    keep[at] = False
    ptab = np.empty(usable_rows, dtype=np.int64)
    ptab[first] = distinct
    ptab[~first] = ts[keep][::-1]                                 # ascending leftovers, handed out from the last repeated row backwards
    return pin, ptab


# ----------------------------------------------------------------------------------------------
# the quotient numerator ([DEP] halo2-axiom plonk/evaluation.rs, reached from
# /root/reference/aggregator/src/wrapper.rs:129).  Two independent restatements:
#   * row_program_run  -- interpreter of the row-program ABI (include/zkhip.h), row by row, big ints;
#   * evaluate_h_direct -- the constraint formulas of `Evaluator::evaluate_h` written out directly (custom gates,
#     permutation argument, lookup argument, Horner in y), without going through a program.
# ----------------------------------------------------------------------------------------------
FR_DELTA = pow(FR_GENERATOR, 1 << FR_S, R_MOD)      # `Fr::DELTA` [DEP ff::PrimeField]: generator of the t-order subgroup


def row_program_run(insns, constants, rotations, rot_scale, result_reg, columns, log_rows, omega=None, prev=None, n_regs=16,
                    only_rows=None):
    """insns: (op, dst, a, b, c) with operands (kind, index, rot_slot); kinds 0 const, 1 reg, 2 column, 3 prev, 4 omega^row;
    ops 0 mov, 1 add, 2 sub, 3 mul, 4 neg, 5 dbl, 6 sqr, 7 mad (a*b + c)."""
    rows = 1 << log_rows
    out = []
    for row in (range(rows) if only_rows is None else only_rows):
        regs = [0] * n_regs
        xp = pow(omega, row, R_MOD) if omega is not None else None

        def val(o):
            kind, index, rot = o
            if kind == 0: return constants[index]
            if kind == 1: return regs[index]
            if kind == 2: return columns[index][(row + rotations[rot] * rot_scale) % rows]
            if kind == 3: return prev[row] if prev is not None else 0
            if kind == 4: return xp
            raise ValueError(kind)

        for op, dst, a, b, c in insns:
            if op == 0: v = val(a)
            elif op == 1: v = val(a) + val(b)
            elif op == 2: v = val(a) - val(b)
            elif op == 3: v = val(a) * val(b)
            elif op == 4: v = -val(a)
            elif op == 5: v = 2 * val(a)
            elif op == 6: v = val(a) ** 2
            elif op == 7: v = val(a) * val(b) + val(c)
            else: raise ValueError(op)
            regs[dst] = v % R_MOD
        out.append(regs[result_reg])
    return out


def eval_expression(e, fixed, advice, instance, challenges, row, rows, rot_scale):
    """`plonk::Expression::evaluate` at one row of the extended coset; e has fields kind / a / b."""
    rec = lambda x: eval_expression(x, fixed, advice, instance, challenges, row, rows, rot_scale)
    k = e.kind
    if k == "constant": return e.a % R_MOD
    if k == "fixed": return fixed[e.a][(row + e.b * rot_scale) % rows]
    if k == "advice": return advice[e.a][(row + e.b * rot_scale) % rows]
    if k == "instance": return instance[e.a][(row + e.b * rot_scale) % rows]
    if k == "challenge": return challenges[e.a]
    if k == "neg": return -rec(e.a) % R_MOD
    if k == "sum": return (rec(e.a) + rec(e.b)) % R_MOD
    if k == "product": return rec(e.a) * rec(e.b) % R_MOD
    if k == "scaled": return rec(e.a) * e.b % R_MOD
    raise ValueError(k)


def evaluate_h_direct(cs, k, extended_k, fixed, advice, instance, l0, l_last, l_active, sigma, perm_products, lookups,
                      beta, gamma, theta, y, challenges=(), zeta=FR_ZETA):
    """`values` of evaluate_h before divide_by_vanishing_poly.  All column arguments are lists of ints over the extended coset;
    lookups: per lookup (product, permuted_input, permuted_table)."""
    rows = 1 << extended_k
    rot_scale = 1 << (extended_k - k)
    omega_ext = omega_for(extended_k)
    last_rotation = -(cs.blinding_factors + 1)
    chunk_len = cs.degree - 2
    pick = {"fixed": fixed, "advice": advice, "instance": instance}
    out = []
    for idx in range(rows):
        r_next = (idx + rot_scale) % rows
        r_prev = (idx - rot_scale) % rows
        r_last = (idx + last_rotation * rot_scale) % rows
        value = 0
        for polys in cs.gates:
            for poly in polys:
                value = (value * y + eval_expression(poly, fixed, advice, instance, challenges, idx, rows, rot_scale)) % R_MOD
        if cs.permutation_columns:
            sets = perm_products
            first, last = sets[0], sets[-1]
            value = (value * y + (1 - first[idx]) * l0[idx]) % R_MOD
            value = (value * y + (last[idx] * last[idx] - last[idx]) * l_last[idx]) % R_MOD
            for si in range(1, len(sets)):
                value = (value * y + (sets[si][idx] - sets[si - 1][r_last]) * l0[idx]) % R_MOD
            current_delta = beta * zeta % R_MOD * pow(omega_ext, idx, R_MOD) % R_MOD
            for si, zs in enumerate(sets):
                chunk = cs.permutation_columns[si * chunk_len:(si + 1) * chunk_len]
                left = zs[r_next]
                for j, (kind, ci) in enumerate(chunk):
                    left = left * (pick[kind][ci][idx] + beta * sigma[si * chunk_len + j][idx] + gamma) % R_MOD
                right = zs[idx]
                for (kind, ci) in chunk:
                    right = right * (pick[kind][ci][idx] + current_delta + gamma) % R_MOD
                    current_delta = current_delta * FR_DELTA % R_MOD
                value = (value * y + (left - right) * l_active[idx]) % R_MOD
        for lk, (prod, pin, ptab) in zip(cs.lookups, lookups):
            cin = 0
            for e in lk.input_expressions:
                cin = (cin * theta + eval_expression(e, fixed, advice, instance, challenges, idx, rows, rot_scale)) % R_MOD
            ctab = 0
            for e in lk.table_expressions:
                ctab = (ctab * theta + eval_expression(e, fixed, advice, instance, challenges, idx, rows, rot_scale)) % R_MOD
            table_value = (cin + beta) * (ctab + gamma) % R_MOD
            a_minus_s = (pin[idx] - ptab[idx]) % R_MOD
            value = (value * y + (1 - prod[idx]) * l0[idx]) % R_MOD
            value = (value * y + (prod[idx] * prod[idx] - prod[idx]) * l_last[idx]) % R_MOD
            value = (value * y + (prod[r_next] * (pin[idx] + beta) * (ptab[idx] + gamma) - prod[idx] * table_value) * l_active[idx]) % R_MOD
            value = (value * y + a_minus_s * l0[idx]) % R_MOD
            value = (value * y + a_minus_s * (pin[idx] - pin[r_prev]) * l_active[idx]) % R_MOD
        out.append(value)
    return out


# ----------------------------------------------------------------------------------------------
# Deterministic input generators shared by tests / bench / C oracle
# ----------------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def fr(self) -> int:
        return from_limbs([self.next() for _ in range(4)]) % R_MOD


def witness_like_scalars(seed: int, n: int) -> List[int]:
    """60 % zero, 30 % uniform < 2^88, 10 % uniform Fr."""
    g = SplitMix64(seed)
    out = []
    for _ in range(n):
        sel = g.next() % 10
        v = g.fr()
        out.append(0 if sel < 6 else (v & ((1 << 88) - 1)) if sel < 9 else v)
    return out


# ----------------------------------------------------------------------------------------------
# Byte/limb encodings of the drop-in boundary (same memory as &[Fr], &[G1Affine], G1)
# ----------------------------------------------------------------------------------------------
def fr_to_limbs(x: int) -> List[int]:
    return limbs4(to_mont(x, R_MOD))


def fr_from_limbs(l: Sequence[int]) -> int:
    return from_mont(from_limbs(l), R_MOD)


def affine_to_limbs(P: Affine) -> List[int]:
    if P is None:
        return [0] * 8
    return limbs4(to_mont(P[0], Q_MOD)) + limbs4(to_mont(P[1], Q_MOD))


def affine_from_limbs(l: Sequence[int]) -> Affine:
    x, y = from_limbs(l[0:4]), from_limbs(l[4:8])
    if x == 0 and y == 0:
        return None
    return (from_mont(x, Q_MOD), from_mont(y, Q_MOD))


def jac_from_limbs(l: Sequence[int]) -> Jac:
    return tuple(from_mont(from_limbs(l[4 * i:4 * i + 4]), Q_MOD) for i in range(3))  # type: ignore
