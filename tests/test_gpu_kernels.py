"""GPU parity tests (kernel level): HIP path through the C ABI vs. the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import oracle_lib

P = 2**64 - 2**32 + 1
pytestmark = pytest.mark.gpu


def rand_field(rng, shape):
    v = rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return np.where(v >= np.uint64(P), v - np.uint64(P), v)


def test_poseidon_permutation_matches_oracle(gpu_ctx, oracle):
    rng = np.random.default_rng(1)
    st = rand_field(rng, (1000, 12))
    st[0] = 0
    st[1] = np.arange(12, dtype=np.uint64)
    st[2] = np.uint64(P - 1)
    got = gpu_ctx.poseidon_permute(st)
    for i in range(st.shape[0]):
        e = st[i].copy()
        oracle.orc_poseidon_permute(oracle_lib.ptr(e))
        assert np.array_equal(got[i], e), i
    assert "%016x" % int(got[0][0]) == "3c18a9786cb0b359"  # upstream KAT


@pytest.mark.parametrize("ncols", [1, 5, 13])
def test_commit_values_matches_oracle(gpu_ctx, oracle, ncols):
    """PolynomialBatch::from_values: coefficients, bit-reversed LDE and Merkle cap, every element."""
    rng = np.random.default_rng(100 + ncols)
    vals = rand_field(rng, (ncols, 65536))
    if ncols > 1:
        vals[1] = 0                       # an all-zero (padding-like) column
        vals[2, :] = np.uint64(P - 1)     # constant column at the top of the range
    c_ref, l_ref, cap_ref = oracle_lib.commit_values(oracle, vals)
    c, l, cap = gpu_ctx.commit_values(vals)
    assert np.array_equal(c, c_ref)
    assert np.array_equal(l, l_ref)
    assert np.array_equal(cap, cap_ref)


def test_commit_values_structured_columns(gpu_ctx, oracle):
    """from_values on columns made of boundary values (0, 1, p-1, 2^32 +- 1, 2^64 - 2^32, ...) in random mixtures, impulses,
    alternating and constant columns: the carry / borrow / wrap paths of the hand-written butterflies, twiddles and products
    inside the real kernels, every coefficient and every LDE value against the oracle."""
    rng = np.random.default_rng(2024)
    edge = np.array([0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, P - 2**32, 2**63, 2**48, 0xFFFFFFFE00000001,
                     0xFFFFFFFEFFFFFFFF, 0x00000001FFFFFFFF, 0xFFFF0000FFFF0001], dtype=np.uint64)
    cols = []
    for k in range(6):
        cols.append(edge[rng.integers(0, edge.size, size=65536)])           # dense mixtures of boundary values
    imp = np.zeros(65536, np.uint64); imp[0] = 1; cols.append(imp)            # impulse: flat spectrum
    imp = np.zeros(65536, np.uint64); imp[65535] = np.uint64(P - 1); cols.append(imp)
    alt = np.zeros(65536, np.uint64); alt[::2] = np.uint64(P - 1); cols.append(alt)
    cols.append(np.full(65536, np.uint64(2**32 - 1)))
    sparse = np.zeros(65536, np.uint64); sparse[rng.integers(0, 65536, size=40)] = edge[rng.integers(0, edge.size, size=40)]
    cols.append(sparse)
    mix = rand_field(rng, 65536); mix[rng.integers(0, 65536, size=20000)] = np.uint64(P - 1); cols.append(mix)
    vals = np.stack(cols)
    c_ref, l_ref, cap_ref = oracle_lib.commit_values(oracle, vals)
    c, l, cap = gpu_ctx.commit_values(vals)
    assert np.array_equal(c, c_ref)
    assert np.array_equal(l, l_ref)
    assert np.array_equal(cap, cap_ref)


def test_ntt_linearity_full_width(gpu_ctx):
    """Size-independent property at the bench's full width: LDE(a+b) = LDE(a)+LDE(b) on 64 columns."""
    rng = np.random.default_rng(7)
    a = rand_field(rng, (32, 65536))
    b = rand_field(rng, (32, 65536))
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    _, la, _ = gpu_ctx.commit_values(a, want_coeffs=False)
    _, lb, _ = gpu_ctx.commit_values(b, want_coeffs=False)
    _, ls, _ = gpu_ctx.commit_values(s, want_coeffs=False)
    assert np.array_equal(((la.astype(object) + lb.astype(object)) % P).astype(np.uint64), ls)


def test_field_asm_edge_cases(gpu_ctx):
    """The hand-written gfx950 sequences of the NTT butterflies (csrc/gl_asm.h): sum, difference, product, multiplication by
    2^S / 2^-K, against Python integers on boundary operands (0, 1, p-1, 2^32 +- 1, 2^64 - 2^32, words of all ones / zeros)
    in every pairing, and on random operands."""
    edge = [0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**63 - 1, P - 2**32, P - 2**32 - 1, P - 2**32 + 1,
            2**64 - 2**33, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFE, 0x0000000100000000,
            0x8000000000000000, 0x7FFFFFFF80000000, 2**48, 2**48 - 1, 2**16, 0xFFFF0000FFFF0001, 0x0000FFFF00000000,
            0x00000001FFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFE, 0x1000, 0xFFF, 0xFFFFFFFF00000000 - 1]
    edge = sorted({e % P for e in edge})
    rng = np.random.default_rng(5)
    a = [x for x in edge for _ in edge] + [int(v) for v in rand_field(rng, 20000)]
    b = [y for _ in edge for y in edge] + [int(v) for v in rand_field(rng, 20000)]
    # sparse operands: few bits set, to reach the carry / borrow corners of the multi-word steps
    for _ in range(20000):
        a.append(sum(1 << int(k) for k in rng.integers(0, 64, size=rng.integers(1, 4))) % P)
        b.append((P - sum(1 << int(k) for k in rng.integers(0, 64, size=rng.integers(1, 4)))) % P)
    out = gpu_ctx.selftest_field(np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64))
    shl = [12, 24, 32, 36, 48, 60, 64, 1, 31]
    shr = [12, 24, 1, 31]
    for i, (x, y) in enumerate(zip(a, b)):
        exp = [(x + y) % P, (x - y) % P, (y - x) % P, x * y % P] + [x * pow(2, s, P) % P for s in shl] + \
              [x * pow(2, 192 - k, P) % P for k in shr]
        got = [int(v) for v in out[i]]
        assert got == exp, (hex(x), hex(y), [j for j in range(17) if got[j] != exp[j]])


def test_fq_inversion_by_divsteps(gpu_ctx):
    """fq_inv (csrc/fq_dev.h: Bernstein-Yang divsteps in 30-bit signed limbs, 20 rounds of 30) against Python's pow(x, -1, p) and
    against the Fermat form on the device: edge values (1, 2, p-1, p-2, (p+1)/2, powers of two, values with long runs of ones
    or zeros in the low limbs, which drive the divstep matrices to their extremes), 0 -> 0, and random elements."""
    from tools import synth
    P = synth.P
    import random
    rng = random.Random(2024)
    xs = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, (P - 1) // 2, 1 << 253, (1 << 253) + 1, (1 << 200) - 1, (1 << 254) % P,
          (1 << 30) - 1, 1 << 30, (1 << 60) + 1, P - (1 << 30), P >> 1, 0x5555555555555555555555555555555555555555555555555555555555555555 % P,
          0x3333333333333333333333333333333333333333333333333333333333333333 % P]
    xs += [(1 << k) % P for k in range(1, 254, 7)] + [P - ((1 << k) % P) for k in range(1, 254, 11)]
    xs += [rng.randrange(1, P) for _ in range(4000)] + [rng.randrange(1, 1 << 64) for _ in range(200)]
    arr = np.array([[(x >> (64 * i)) & synth.MASK64 for i in range(4)] for x in xs], dtype=np.uint64)
    out = gpu_ctx.selftest_fq_inv(arr)
    for x, row in zip(xs, out):
        got = synth.words_to_int(row[:4])
        want = pow(x, -1, P) if x else 0
        assert got == want and synth.words_to_int(row[4:]) == want, hex(x)
