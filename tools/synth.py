"""Deterministic synthetic inputs for the BN254 scalar-mul STARKs (SURVEY.md §8(d) "Synthetic inputs").

Pure-Python big-integer BN254 arithmetic: independent of both the HIP build and the C++ oracle, so it
also serves as the generator of golden vectors for the BN254 layer (tools/gen_bn254_golden.py).
Mirrors the reference's test inputs (src/starks/curves/g1/scalar_mul_stark.rs:557-566): scalar = 32
uniformly random bytes (not reduced mod r), x and offset = random non-infinity G1 points.
"""
from __future__ import annotations

import numpy as np

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
G1_GEN = (1, 2)
MASK64 = (1 << 64) - 1


class Xoshiro256ss:
    """xoshiro256** seeded through splitmix64 (public-domain algorithm by Blackman & Vigna)."""

    def __init__(self, seed: int):
        s = seed & MASK64
        self.s = []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & MASK64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
            self.s.append(z ^ (z >> 31))

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK64

    def next_u64(self) -> int:
        s = self.s
        result = (self._rotl((s[1] * 5) & MASK64, 7) * 9) & MASK64
        t = (s[1] << 17) & MASK64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def next_u256(self) -> int:
        return sum(self.next_u64() << (64 * i) for i in range(4))


# ---- G1 affine / Jacobian arithmetic over Python ints -------------------------------------------------
def g1_add(a, b):
    """Affine add of non-infinity points with b != -a (textbook short-Weierstrass formulas)."""
    (x1, y1), (x2, y2) = a, b
    if x1 != x2:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    else:
        if (y1 + y2) % P == 0:
            raise ValueError("point at infinity")
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def _jac_double(p):
    x, y, z = p
    if y == 0:
        return (1, 1, 0)
    a = x * x % P
    b = y * y % P
    c = b * b % P
    d = 2 * ((x + b) * (x + b) - a - c) % P
    e = 3 * a % P
    f = e * e % P
    x3 = (f - 2 * d) % P
    y3 = (e * (d - x3) - 8 * c) % P
    z3 = 2 * y * z % P
    return (x3, y3, z3)


def _jac_add_affine(p, q):
    x1, y1, z1 = p
    x2, y2 = q
    if z1 == 0:
        return (x2, y2, 1)
    z1z1 = z1 * z1 % P
    u2 = x2 * z1z1 % P
    s2 = y2 * z1 * z1z1 % P
    if u2 == x1:
        if s2 == y1:
            return _jac_double(p)
        return (1, 1, 0)
    h = (u2 - x1) % P
    hh = h * h % P
    i = 4 * hh % P
    j = h * i % P
    r = 2 * (s2 - y1) % P
    v = x1 * i % P
    x3 = (r * r - j - 2 * v) % P
    y3 = (r * (v - x3) - 2 * y1 * j) % P
    z3 = ((z1 + h) * (z1 + h) - z1z1 - hh) % P
    return (x3, y3, z3)


def g1_mul(k: int, pt):
    """k * pt for k >= 1 (returns affine; raises on infinity)."""
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_double(acc)
        if bit == "1":
            acc = _jac_add_affine(acc, pt)
    x, y, z = acc
    if z == 0:
        raise ValueError("point at infinity")
    zi = pow(z, -1, P)
    return (x * zi * zi % P, y * zi * zi * zi % P)


def g1_scalar_mul_offset(s: int, x, offset):
    """s*x + offset as the reference computes the expected output (scalar_mul_stark.rs:105-106)."""
    k = s % R_ORDER
    if k == 0:
        return offset
    return g1_add(g1_mul(k, x), offset)


def _to_words(v: int, n: int = 4):
    return [(v >> (64 * i)) & MASK64 for i in range(n)]


def g1_inputs(n: int, seed: int = 0x706C6F6E6B7932):
    """n synthetic G1 scalar-mul jobs in ABI wire form.

    Returns (scalars[n,4], x[n,8], offset[n,8]) as uint64 numpy arrays: little-endian 64-bit words,
    canonical (non-Montgomery) coordinates, x then y.
    """
    rng = Xoshiro256ss(seed)
    scalars = np.zeros((n, 4), dtype=np.uint64)
    xs = np.zeros((n, 8), dtype=np.uint64)
    offs = np.zeros((n, 8), dtype=np.uint64)
    for i in range(n):
        s = rng.next_u256()
        k1 = rng.next_u256() % (R_ORDER - 1) + 1
        k2 = rng.next_u256() % (R_ORDER - 1) + 1
        x = g1_mul(k1, G1_GEN)
        off = g1_mul(k2, G1_GEN)
        scalars[i] = _to_words(s)
        xs[i] = _to_words(x[0]) + _to_words(x[1])
        offs[i] = _to_words(off[0]) + _to_words(off[1])
    return scalars, xs, offs


def words_to_int(w) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(w))


# ---- Fq2 = Fq[u]/(u^2+1) and G2 (y^2 = x^3 + b2) over Python ints ---------------------------------------
G2_B = (19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690)   # src/curves/g2.rs:29-36
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def g2_add(a, b):
    """Affine add on the twist (textbook formulas), b != -a."""
    (x1, y1), (x2, y2) = a, b
    if x1 != x2:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    else:
        if f2_add(y1, y2) == (0, 0):
            raise ValueError("point at infinity")
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_mul((2, 0), y1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(k: int, pt):
    """k * pt (k >= 1) by affine double-and-add; raises on infinity."""
    acc = None
    for bit in bin(k)[2:]:
        if acc is not None:
            acc = g2_add(acc, acc)
        if bit == "1":
            acc = pt if acc is None else g2_add(acc, pt)
    return acc


def g2_scalar_mul_offset(s: int, x, offset):
    k = s % R_ORDER
    if k == 0:
        return offset
    return g2_add(g2_mul(k, x), offset)


def g2_inputs(n: int, seed: int = 0x706C6F6E6B7932 + 3):
    """(scalars[n,4], x[n,16], offset[n,16]); point = x.c0, x.c1, y.c0, y.c1 (4 words each)."""
    rng = Xoshiro256ss(seed)
    scalars = np.zeros((n, 4), dtype=np.uint64)
    xs = np.zeros((n, 16), dtype=np.uint64)
    offs = np.zeros((n, 16), dtype=np.uint64)
    for i in range(n):
        s = rng.next_u256()
        k1 = rng.next_u256() % (R_ORDER - 1) + 1
        k2 = rng.next_u256() % (R_ORDER - 1) + 1
        x = g2_mul(k1, G2_GEN)
        off = g2_mul(k2, G2_GEN)
        scalars[i] = _to_words(s)
        xs[i] = _to_words(x[0][0]) + _to_words(x[0][1]) + _to_words(x[1][0]) + _to_words(x[1][1])
        offs[i] = _to_words(off[0][0]) + _to_words(off[0][1]) + _to_words(off[1][0]) + _to_words(off[1][1])
    return scalars, xs, offs


def g2_from_words(w):
    return ((words_to_int(w[0:4]), words_to_int(w[4:8])), (words_to_int(w[8:12]), words_to_int(w[12:16])))


def fq_inputs(n: int, seed: int = 0x706C6F6E6B7932 + 5):
    """(scalars[n,4], x[n,4]) for the Fq exponentiation STARK: x uniform in [0,p), s any 256-bit value."""
    rng = Xoshiro256ss(seed)
    scalars = np.zeros((n, 4), dtype=np.uint64)
    xs = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        scalars[i] = _to_words(rng.next_u256())
        xs[i] = _to_words(rng.next_u256() % P)
    return scalars, xs
