"""GPU parity: G1 trace generation (HIP) vs. the CPU oracle restatement of generate_trace, every cell."""
import numpy as np
import pytest

from tools import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def oracle_trace(oracle, s, x, o, min_rows_log2):
    n = s.shape[0]
    rows = oracle.orc_g1_num_rows(n, min_rows_log2)
    tr = np.zeros((781, rows), np.uint64)
    outs = np.zeros((n, 8), np.uint64)
    rc = oracle.orc_g1_generate_trace(oracle_lib.ptr(s), oracle_lib.ptr(x), oracle_lib.ptr(o), n, min_rows_log2,
                                      oracle_lib.ptr(tr), oracle_lib.ptr(outs))
    assert rc == 0, oracle.orc_last_error()
    return tr, outs


def edge_inputs():
    """scalar 0, scalar 2^256-1, x == offset (first add is a doubling), one-bit scalars, plus random."""
    s, x, o = synth.g1_inputs(6, seed=99)
    s[0] = 0
    s[1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    o[2] = x[2]            # a == b on an addition row: is_x_eq branch of generate_g1_add (add.rs:75-91)
    s[2, 0] |= np.uint64(1)
    s[3] = 0
    s[3, 3] = np.uint64(1 << 63)   # only the top bit
    s[4] = 0
    s[4, 0] = np.uint64(1)
    return s, x, o


def test_trace_small_with_padding_and_edges(gpu_ctx, oracle):
    s, x, o = edge_inputs()
    ref, ref_out = oracle_trace(oracle, s, x, o, 16)   # 6*512 = 3072 real rows, the rest is padding
    got, got_out = gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
    assert got.shape == ref.shape
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first mismatches (col,row): {bad[:10].tolist()}"
    assert np.array_equal(got_out, ref_out)
    for i in range(6):   # outputs against independent Python big-int arithmetic
        exp = synth.g1_scalar_mul_offset(synth.words_to_int(s[i]),
                                         (synth.words_to_int(x[i, :4]), synth.words_to_int(x[i, 4:])),
                                         (synth.words_to_int(o[i, :4]), synth.words_to_int(o[i, 4:])))
        assert (synth.words_to_int(got_out[i, :4]), synth.words_to_int(got_out[i, 4:])) == exp


def test_trace_full_shape(gpu_ctx, oracle):
    """The reference test shape: 128 instances, 2^16 rows (scalar_mul_stark.rs:554,569)."""
    s, x, o = synth.g1_inputs(128)
    ref, ref_out = oracle_trace(oracle, s, x, o, 16)
    got, got_out = gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first mismatches (col,row): {bad[:10].tolist()}"
    assert np.array_equal(got_out, ref_out)


def test_trace_rejects_opposite_points(gpu_ctx):
    """x == -offset makes the first addition hit the point at infinity: the reference cannot prove it
    (add.rs:49-51); the ABI reports BN254S_E_INVALID_POINT instead of panicking."""
    s, x, o = synth.g1_inputs(2, seed=5)
    py = synth.words_to_int(x[1, 4:])
    neg = synth.P - py
    o[1, :4] = x[1, :4]
    o[1, 4:] = [(neg >> (64 * i)) & synth.MASK64 for i in range(4)]
    with pytest.raises(RuntimeError, match="-4"):
        gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
