"""GPU parity for tall traces (more than 128 calls in ONE proof, as `Bn254Hook::constrain` produces for a circuit with many
calls: reference src/hook.rs:63-71, rows = (512 n).next_power_of_two(), scalar_mul_stark.rs:60): N = 2^17 and 2^18."""
import numpy as np
import pytest

from plonky2_bn254_amd import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def test_fq_exp_tall_proof_2pow18(gpu_ctx, oracle):
    s, x = synth.fq_inputs(300, seed=31)            # 300 * 512 = 153600 rows -> N = 2^18, FRI arities [4,4,4,4]
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 2, s, x)
    assert degree_bits == 18
    pr = gpu_ctx.prove_fq_exp(s, x)
    assert pr.degree_bits == 18 and pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing words {bad[:5]} of {ref.size}"
    assert np.array_equal(pr.outputs.reshape(-1, 4), ref_out)
    rc, msg = oracle_lib.verify(oracle, 2, pr.words, degree_bits, s, x)
    assert rc == 0, msg


def test_g1_tall_proof_2pow17(gpu_ctx, oracle):
    s, x, o = synth.g1_inputs(150, seed=32)         # 76800 rows -> N = 2^17
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 0, s, x, o)
    assert degree_bits == 17
    pr = gpu_ctx.prove_g1(s, x, o)
    assert pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing words {bad[:5]} of {ref.size}"
    rc, msg = oracle_lib.verify(oracle, 0, pr.words, degree_bits, s, x, o)
    assert rc == 0, msg
    print("tall G1 (2^17) stage ms:", {k: round(v, 2) for k, v in pr.stage_ms.items()})
