"""GPU parity for the G2 scalar-mul STARK and the Fq-exp STARK (BASELINE configs 3 and 5): trace cell-by-cell and
the whole proof transcript vs. the CPU oracle, outputs vs. independent Python big-int arithmetic, acceptance by the
restated native verifier."""
import numpy as np
import pytest

from tools import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def g2_edge_inputs(n=5):
    s, x, o = synth.g2_inputs(n)
    s[0] = 0
    o[1] = x[1]                       # a == b on an addition row (doubling branch, is_x_eq = 1)
    s[1, 0] |= np.uint64(1)
    s[2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return s, x, o


def fq_edge_inputs(n=6):
    s, x = synth.fq_inputs(n)
    s[0] = 0                          # x^0 = 1
    x[1] = 0                          # 0^s
    x[1, 0] = 0
    s[2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    x[3] = 0
    x[3, 0] = 1                       # 1^s
    return s, x


def test_fq_arithmetic_limb_boundaries(gpu_ctx, oracle):
    """Field elements that sit on the carry boundaries of the 26-bit-limb representation of csrc/fq_dev.h (all-ones limbs,
    p - 1, p - 2^k, 2^(26 j) +- 1, R mod p ...): the whole Fq-exp trace (which squares and multiplies them 256 times) must
    equal the oracle's, and the outputs Python's pow()."""
    P = synth.P
    vals = [P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (1 << 253) + 12345, (1 << 254) % P, pow(2, 260, P), pow(2, 520, P), P - (1 << 26),
            P - (1 << 52) + 1]
    vals += [(1 << (26 * j)) - 1 for j in range(1, 10)] + [(1 << (26 * j)) + 1 for j in range(1, 10)] + [(1 << (26 * j)) % P for j in (9, 10)]
    vals += [sum(((1 << 26) - 1) << (26 * j) for j in range(9)), sum(0x2AAAAAA << (26 * j) for j in range(9)) % P]
    n = len(vals)
    s, x = synth.fq_inputs(n, seed=77)
    for i, v in enumerate(vals):
        x[i] = synth._to_words(v % P)
    s[1] = synth._to_words(P - 2)          # inversion by exponentiation
    s[2] = synth._to_words((1 << 256) - 1)
    ref, ref_out = oracle_lib.generate_trace(oracle, 2, s, x)
    got, got_out = gpu_ctx.generate_trace(2, s, x)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first mismatches (col,row): {bad[:10].tolist()}"
    for i in range(n):
        assert synth.words_to_int(got_out[i]) == pow(synth.words_to_int(x[i]), synth.words_to_int(s[i]), P), i


def test_g2_trace_matches_oracle(gpu_ctx, oracle):
    s, x, o = g2_edge_inputs()
    ref, ref_out = oracle_lib.generate_trace(oracle, 1, s, x, o)
    got, got_out = gpu_ctx.generate_trace(1, s, x, o)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first mismatches (col,row): {bad[:10].tolist()}"
    assert np.array_equal(got_out, ref_out)
    for i in range(s.shape[0]):
        exp = synth.g2_scalar_mul_offset(synth.words_to_int(s[i]), synth.g2_from_words(x[i]), synth.g2_from_words(o[i]))
        assert synth.g2_from_words(got_out[i]) == exp


def test_fq_trace_matches_oracle(gpu_ctx, oracle):
    s, x = fq_edge_inputs()
    ref, ref_out = oracle_lib.generate_trace(oracle, 2, s, x)
    got, got_out = gpu_ctx.generate_trace(2, s, x)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"first mismatches (col,row): {bad[:10].tolist()}"
    assert np.array_equal(got_out, ref_out)
    for i in range(s.shape[0]):
        assert synth.words_to_int(got_out[i]) == pow(synth.words_to_int(x[i]), synth.words_to_int(s[i]), synth.P)


def test_fq_exp_proof_bit_exact_and_verifies(gpu_ctx, oracle):
    """128 exponentiations, 2^16 rows: the reference's fq_exp test shape (exp_stark.rs:533-600)."""
    s, x = synth.fq_inputs(128)
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 2, s, x)
    pr = gpu_ctx.prove_fq_exp(s, x)
    assert pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing word {bad[:5]}"
    assert np.array_equal(pr.outputs.reshape(-1, 4), ref_out)
    rc, msg = oracle_lib.verify(oracle, 2, pr.words, degree_bits, s, x)
    assert rc == 0, msg
    bad_proof = pr.words.copy()
    bad_proof[300] ^= np.uint64(1)
    assert oracle_lib.verify(oracle, 2, bad_proof, degree_bits, s, x)[0] == 1


def test_g2_proof_bit_exact_and_verifies(gpu_ctx, oracle):
    """G2 scalar-mul STARK (W = 1295): edge-case instances + padding rows, like the reference's 1-input g2 test."""
    s, x, o = g2_edge_inputs()
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 1, s, x, o)
    pr = gpu_ctx.prove_g2(s, x, o)
    assert pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing word {bad[:5]}"
    assert np.array_equal(pr.outputs.reshape(-1, 16), ref_out)
    rc, msg = oracle_lib.verify(oracle, 1, pr.words, degree_bits, s, x, o)
    assert rc == 0, msg


def test_g2_full_batch_verifies(gpu_ctx, oracle):
    """128 G2 instances (a full 2^16-row proof): accepted by the native verifier, outputs = s*x+offset."""
    s, x, o = synth.g2_inputs(128, seed=4242)
    pr = gpu_ctx.prove_g2(s, x, o)
    rc, msg = oracle_lib.verify(oracle, 1, pr.words, pr.degree_bits, s, x, o)
    assert rc == 0, msg
    for i in (0, 77, 127):
        exp = synth.g2_scalar_mul_offset(synth.words_to_int(s[i]), synth.g2_from_words(x[i]), synth.g2_from_words(o[i]))
        assert synth.g2_from_words(pr.outputs.reshape(-1, 16)[i]) == exp
    print("G2 stage ms:", {k: round(v, 2) for k, v in pr.stage_ms.items()})


def test_batches_of_g2_and_fq(gpu_ctx, oracle):
    """BASELINE configs 3/5 shape in miniature: jobs cut into 128-instance proofs, every proof verifies against its slice."""
    s, x, o = synth.g2_inputs(130, seed=9)
    proofs = gpu_ctx.prove_batch(1, s, x, o)
    assert len(proofs) == 2
    for i, pr in enumerate(proofs):
        sl = slice(128 * i, min(128 * (i + 1), 130))
        rc, msg = oracle_lib.verify(oracle, 1, pr.words, pr.degree_bits, s[sl], x[sl], o[sl])
        assert rc == 0, (i, msg)
    s, x = synth.fq_inputs(300, seed=10)
    proofs = gpu_ctx.prove_batch(2, s, x)
    assert len(proofs) == 3
    for i, pr in enumerate(proofs):
        sl = slice(128 * i, min(128 * (i + 1), 300))
        rc, msg = oracle_lib.verify(oracle, 2, pr.words, pr.degree_bits, s[sl], x[sl])
        assert rc == 0, (i, msg)
