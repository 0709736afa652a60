"""Host front-end of the `map_to_g2` pipeline: turns Fq2 inputs into the STARK jobs that BASELINE.json configs[4] proves.

The reference's circuit (`map_to_g2_circuit`, src/utils/hash_to_g2.rs:150-207) maps u in Fq2 to G2 with the
Shallue-van de Woestijne method (RFC 9380 6.6.1, Z = 1) and spends its STARK work in two places:

  * `is_square` of g(x1) and g(x2): the Legendre symbol of the Fq2 norm, i.e. ONE fq_exp job per candidate with exponent
    (p-1)/2 (src/fields/fq2.rs:235-240 -> src/fields/fq.rs:290-292) -> 2 fq_exp jobs per input;
  * clearing the cofactor: `g2_scalar_mul(cofactor, (x, y), offset)` with a random non-infinity offset that is subtracted
    afterwards (hash_to_g2.rs:190-205) -> 1 G2 scalar-mul job per input.

Everything else (a handful of Fq2 products, one square root with a sign) is plain witness arithmetic; here it runs on
Python integers.  `fq_exp_jobs` / `g2_jobs` return arrays in the wire format of include/bn254_stark.h, so config 5 is
64 Fq-exp proofs + 32 G2 proofs for 4096 inputs (sizes table).

Square roots follow ark-ff 0.4 (the crate the reference calls, un-vendored): Fq sqrt = a^((p+1)/4); Fq2 sqrt = the
"complex method" of QuadExtField::sqrt.  Only the fixed constant tv4 depends on which of the two roots that returns;
the sign of y is fixed by sgn(u) (hash_to_g2.rs:141-143, src/fields/sgn.rs:20-27).
"""
from __future__ import annotations

import numpy as np

from tools import synth
from tools.synth import P, f2_add, f2_inv, f2_mul, f2_sub

COFACTOR = 21888242871839275222246405745257275088844257914179612981679871602714643921549  # hash_to_g2.rs:69-71
LEGENDRE_EXP = (P - 1) // 2
ONE, ZERO = (1, 0), (0, 0)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_norm(a):
    return (a[0] * a[0] + a[1] * a[1]) % P


def fq_is_square(a: int) -> bool:
    return a % P == 0 or pow(a, LEGENDRE_EXP, P) == 1


def fq_sqrt(a: int):
    """p = 3 mod 4: a^((p+1)/4), None when a is not a square."""
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


def f2_sqrt(a):
    """QuadExtField::sqrt of ark-ff 0.4 (complex method, eprint 2012/685 alg. 8); None when a is not a square."""
    c0, c1 = a
    if c1 == 0:
        if fq_is_square(c0):
            return (fq_sqrt(c0), 0)
        r = fq_sqrt((-c0) % P)  # c0 / nonresidue, nonresidue = -1
        return None if r is None else (0, r)
    alpha = fq_sqrt(f2_norm(a))
    if alpha is None:
        return None
    two_inv = (P + 1) // 2
    delta = (alpha + c0) * two_inv % P
    if not fq_is_square(delta):
        delta = (delta - alpha) % P
    r0 = fq_sqrt(delta)
    if r0 is None or r0 == 0:
        return None
    cand = (r0, c1 * two_inv % P * pow(r0, -1, P) % P)
    return cand if f2_mul(cand, cand) == (c0 % P, c1 % P) else None


def g(x):
    """Right-hand side of the twist equation y^2 = x^3 + b2 (src/curves/g2.rs:29-36)."""
    return f2_add(f2_mul(f2_mul(x, x), x), synth.G2_B)


def sgn(a) -> bool:
    """src/fields/sgn.rs:20-27: parity of c0, or of c1 when c0 = 0."""
    return bool(a[0] & 1) or (a[0] == 0 and bool(a[1] & 1))


# constants of map_to_g2 for Z = 1 (hash_to_g2.rs:114-118)
_Z = ONE
_GZ = g(_Z)
_NEG_Z_BY_2 = f2_mul(f2_neg(_Z), f2_inv((2, 0)))
_TV4 = f2_sqrt(f2_mul(f2_neg(_GZ), (3, 0)))
_TV6 = f2_mul(f2_mul((P - 4, 0), _GZ), f2_inv((3, 0)))
assert _TV4 is not None


def candidates(u):
    """(x1, x2, x3) of hash_to_g2.rs:119-127."""
    tv1 = f2_mul(f2_mul(u, u), _GZ)
    tv2 = f2_add(ONE, tv1)
    tv1 = f2_sub(ONE, tv1)
    tv3 = f2_inv(f2_mul(tv1, tv2))
    tv5 = f2_mul(f2_mul(f2_mul(u, tv1), tv3), _TV4)
    x1 = f2_sub(_NEG_Z_BY_2, tv5)
    x2 = f2_add(_NEG_Z_BY_2, tv5)
    t = f2_mul(f2_mul(tv2, tv2), tv3)
    x3 = f2_add(_Z, f2_mul(_TV6, f2_mul(t, t)))
    return x1, x2, x3


def fq_exp_jobs(us):
    """The 2n Legendre jobs (exponent (p-1)/2, base norm(g(x_i))) as (scalars[2n,4], x[2n,4]); job 2k+i belongs to
    candidate x_{i+1} of input k."""
    n = len(us)
    scalars = np.zeros((2 * n, 4), np.uint64)
    xs = np.zeros((2 * n, 4), np.uint64)
    for k, u in enumerate(us):
        x1, x2, _ = candidates(u)
        for i, xc in enumerate((x1, x2)):
            scalars[2 * k + i] = synth._to_words(LEGENDRE_EXP)
            xs[2 * k + i] = synth._to_words(f2_norm(g(xc)))
    return scalars, xs


def select_point(u, is_gx1_sq: bool, is_gx2_sq: bool):
    """(x, y) on the twist before cofactor clearing, given the two Legendre results (hash_to_g2.rs:128-145)."""
    x1, x2, x3 = candidates(u)
    x = x1 if is_gx1_sq else x2 if is_gx2_sq else x3
    y = f2_sqrt(g(x))
    if y is None:
        raise ValueError("g(x) is not a square: inconsistent Legendre results")
    if sgn(u) != sgn(y):
        y = f2_neg(y)
    return x, y


def _pt_words(pt):
    return synth._to_words(pt[0][0]) + synth._to_words(pt[0][1]) + synth._to_words(pt[1][0]) + synth._to_words(pt[1][1])


def g2_jobs(us, legendre_outputs, seed: int = 0x706C6F6E6B7932 + 5):
    """The n cofactor-clearing jobs (scalars[n,4], x[n,16], offset[n,16]).  legendre_outputs[2k+i] = the fq_exp result of
    job 2k+i as an integer (1 <=> square; 0 for the zero norm also counts as square, as ark's `legendre().is_qr()` does
    not: a zero norm means g(x) = 0, which no random input reaches).  Offsets are random subgroup points (set_random_g2)."""
    rng = synth.Xoshiro256ss(seed)
    n = len(us)
    scalars = np.zeros((n, 4), np.uint64)
    xs = np.zeros((n, 16), np.uint64)
    offs = np.zeros((n, 16), np.uint64)
    pts = []
    for k, u in enumerate(us):
        pt = select_point(u, int(legendre_outputs[2 * k]) == 1, int(legendre_outputs[2 * k + 1]) == 1)
        off = synth.g2_mul(rng.next_u256() % (synth.R_ORDER - 1) + 1, synth.G2_GEN)
        scalars[k] = synth._to_words(COFACTOR)
        xs[k] = _pt_words(pt)
        offs[k] = _pt_words(off)
        pts.append((pt, off))
    return scalars, xs, offs, pts


def finish(output_words, offset_pt):
    """output - offset: the point map_to_g2 returns (hash_to_g2.rs:200-205)."""
    out = synth.g2_from_words(output_words)
    neg = (offset_pt[0], f2_neg(offset_pt[1]))
    return synth.g2_add(out, neg)


def inputs(n: int, seed: int = 0x706C6F6E6B7932 + 5):
    """n uniform Fq2 inputs u."""
    rng = synth.Xoshiro256ss(seed)
    return [(rng.next_u256() % P, rng.next_u256() % P) for _ in range(n)]


# ---- hash_to_fq2 (hash_to_g2.rs:76-87): pure-Python mirror of bn254s_hash_to_fq2 -------------------------------------------
GL_P = 0xFFFFFFFF00000001
_MDS = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]


def _round_constants():
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonky2_bn254_amd", "csrc", "poseidon_constants.inc")
    return [int(h, 16) for h in re.findall(r"0x([0-9a-fA-F]{16})ULL", open(path).read())]


def poseidon_permute(state, rc=None):
    rc = rc or _round_constants()
    s = list(state)
    for rnd in range(30):
        s = [(s[i] + rc[12 * rnd + i]) % GL_P for i in range(12)]
        if rnd < 4 or rnd >= 26:
            s = [pow(v, 7, GL_P) for v in s]
        else:
            s[0] = pow(s[0], 7, GL_P)
        s = [(sum(s[(i + r) % 12] * _MDS[i] for i in range(12)) + (8 * s[0] if r == 0 else 0)) % GL_P for r in range(12)]
    return s


def hash_to_fq2(inputs):
    """plonky2 Challenger (overwrite-mode duplex sponge, rate 8): observe the inputs, draw 2 x 16 challenges; a coordinate
    is the little-endian 512-bit integer of the challenges' low 32 bits, modulo p."""
    rc = _round_constants()
    state, buf, out = [0] * 12, [], []

    def duplex():
        nonlocal state, buf, out
        for i, v in enumerate(buf):
            state[i] = v
        buf = []
        state = poseidon_permute(state, rc)
        out = state[:8]

    for v in inputs:
        out = []
        buf.append(int(v) % GL_P)
        if len(buf) == 8:
            duplex()

    def challenge():
        if buf or not out:
            duplex()
        return out.pop()

    res = []
    for _ in range(2):
        limbs = [challenge() & 0xFFFFFFFF for _ in range(16)]
        res.append(sum(v << (32 * i) for i, v in enumerate(limbs)) % P)
    return tuple(res)
