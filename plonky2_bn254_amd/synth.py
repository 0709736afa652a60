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
