"""GPU proofs against committed golden digests (tests/golden/proof_digests.json, made by tools/gen_proof_golden.py from the
CPU oracle on fixed synthetic inputs): the whole transcript, the three caps, the outputs and the proof-of-work witness,
without running the oracle at test time."""
import hashlib
import json
import os

import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "proof_digests.json")
GEN = {0: synth.g1_inputs, 1: synth.g2_inputs, 2: synth.fq_inputs}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


# g1_tall19: BASELINE configs[1] (1024 scalar multiplications) as ONE proof of 2^19 rows - what Bn254Hook::constrain
# (hook.rs:63-71) produces for a circuit with 1024 calls; the digest was made by the oracle on a 128-thread host (2.3 min)
@pytest.mark.parametrize("name", ["g1", "g1_full", "g2", "fq_exp", "g1_tall19"])
def test_proof_matches_golden_digest(gpu_ctx, name):
    g = json.load(open(GOLDEN))[name]
    ins = GEN[g["kind"]](g["n"], seed=g["seed"])
    off = ins[2] if len(ins) > 2 else None
    pr = gpu_ctx.prove_batch(g["kind"], ins[0], ins[1], off, per_proof=g["n"])[0]
    assert pr.degree_bits == g["degree_bits"] and pr.words.size == g["n_words"]
    assert [int(w) for w in pr.words[:4]] == g["first_words"]
    assert digest(pr.words[:64]) == g["sha256_trace_cap"]
    assert digest(pr.words[64:128]) == g["sha256_aux_cap"]
    assert digest(pr.words[128:192]) == g["sha256_quotient_cap"]
    assert int(pr.words[-13]) == g["pow_witness"]
    assert digest(pr.outputs) == g["sha256_outputs"]
    assert digest(pr.words) == g["sha256_proof_words"]
