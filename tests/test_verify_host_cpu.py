"""bn254s_verify_host on the CPU (no GPU, no context): the library's GPU-free verifier - transcript replay, the AIR evaluated
at zeta over the quadratic extension by csrc/verify_air_host.h, FRI, Merkle paths, CTL sums - against a proof made by the CPU
oracle, mirroring the reference's verify() (src/starks/common/verifier.rs:32-98) right after prove() in run_once
(src/generators/fq/stark_proof.rs:163-171).  The G1 / G2 AIRs go through the same function in tests/test_gpu_verify.py."""
import numpy as np
import pytest

import plonky2_bn254_amd as pk
from tools import synth
from tests import oracle_lib


@pytest.fixture(scope="module")
def fq_proof(oracle):
    s, x = synth.fq_inputs(3, seed=0xA4)
    words, outs, _, degree_bits = oracle_lib.prove(oracle, 2, s, x)
    return s, x, words, outs, degree_bits


def test_host_verifier_accepts_the_oracle_proof(fq_proof):
    s, x, words, outs, degree_bits = fq_proof
    pk.verify_host(2, words, degree_bits, s, x, None, outs)


@pytest.mark.parametrize("where,expect", [
    (192 + 7, "Mismatch between evaluation and opening of quotient polynomial"),          # a local_values opening
    (192 + 4 * 427 + 3, "Mismatch between evaluation and opening of quotient polynomial"),  # an auxiliary_polys opening
    (5, "init_challenger_state mismatch"),                                                 # the trace cap
])
def test_host_verifier_rejects_corruption(fq_proof, where, expect):
    s, x, words, outs, degree_bits = fq_proof
    bad = words.copy()
    bad[where] ^= np.uint64(1)
    with pytest.raises(pk.VerifyError, match=expect):
        pk.verify_host(2, bad, degree_bits, s, x, None, outs)


def test_host_verifier_checks_the_claimed_inputs_and_outputs(fq_proof):
    s, x, words, outs, degree_bits = fq_proof
    wrong = outs.copy()
    wrong[0, 0] ^= np.uint64(1)
    with pytest.raises(pk.VerifyError, match="CTL sum mismatch"):
        pk.verify_host(2, words, degree_bits, s, x, None, wrong)
    s2 = s.copy()
    s2[1, 0] ^= np.uint64(2)
    with pytest.raises(pk.VerifyError, match="CTL sum mismatch"):
        pk.verify_host(2, words, degree_bits, s2, x, None, outs)
    with pytest.raises(pk.VerifyError, match="bad proof shape"):
        pk.verify_host(2, words[:-1], degree_bits, s, x, None, outs)
