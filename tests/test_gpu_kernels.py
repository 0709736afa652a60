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
