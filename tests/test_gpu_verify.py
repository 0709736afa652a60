"""bn254s_verify (the product's own verifier: csrc/verify.hip) on GPU proofs of the three STARKs.

Mirrors what the reference does right after proving (src/generators/g1/stark_proof.rs:164-172: native `verify` with the
extra looking CTL values) and its negative space: every section of the proof is corrupted in turn and must be rejected
with the reference verifier's error text; wrong claimed inputs / outputs fail the CTL sum check (ctl_values.rs:28-47).
The CPU oracle's restated verifier must agree on every case (accept and reject).
"""
import numpy as np
import pytest

import plonky2_bn254_amd as pk
from tools import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu

SHAPES = {0: (781, 456), 1: (1295, 906), 2: (427, 134)}


def sections(kind, n_words):
    """name -> (first word, length) of the layout documented in include/bn254_stark.h (2^16-row proof: 3 FRI layers)."""
    W, A = SHAPES[kind]
    out, pos = {}, 0
    for name, ln in (("trace_cap", 64), ("aux_cap", 64), ("quotient_cap", 64), ("local_values", 2 * W), ("next_values", 2 * W),
                     ("auxiliary_polys", 2 * A), ("auxiliary_polys_next", 2 * A), ("ctl_zs_first", 4), ("quotient_polys", 8),
                     ("fri_caps", 3 * 64)):
        out[name] = (pos, ln)
        pos += ln
    tail = 2 * 16 + 1 + 12  # final_poly (16 extension coefficients), pow_witness, init_challenger_state
    out["queries"] = (pos, n_words - tail - pos)
    out["final_poly"] = (n_words - tail, 32)
    out["pow_witness"] = (n_words - 13, 1)
    out["init_state"] = (n_words - 12, 12)
    return out


@pytest.fixture(scope="module")
def g1(gpu_ctx):
    s, x, o = synth.g1_inputs(128)
    pr = gpu_ctx.prove_g1(s, x, o)
    return dict(kind=0, s=s, x=x, o=o, words=pr.words.copy(), outputs=pr.outputs.copy(), degree_bits=pr.degree_bits)


def both_verify(gpu_ctx, oracle, p, words=None, s=None, x=None, o=None, outputs=None):
    """(product verdict, oracle verdict): None when accepted, else the error text."""
    words = p["words"] if words is None else words
    s = p["s"] if s is None else s
    x = p["x"] if x is None else x
    o = p["o"] if o is None else o
    outputs = p["outputs"] if outputs is None else outputs
    try:
        gpu_ctx.verify(p["kind"], words, p["degree_bits"], s, x, o, outputs)
        mine = None
    except pk.VerifyError as e:
        mine = str(e)
    # the GPU-free form (bn254s_verify_host: the AIR evaluated on the host over F2, an independent statement) must give the
    # same verdict and the same text on every case of this file
    try:
        pk.verify_host(p["kind"], words, p["degree_bits"], s, x, o, outputs)
        host = None
    except pk.VerifyError as e:
        host = str(e)
    assert host == mine, (host, mine)
    rc, msg = oracle_lib.verify(oracle, p["kind"], words, p["degree_bits"], s, x, o)
    return mine, (None if rc == 0 else msg)


def test_accepts_g1_proof(gpu_ctx, oracle, g1):
    mine, ref = both_verify(gpu_ctx, oracle, g1)
    assert mine is None and ref is None, (mine, ref)


# (most `next` openings enter no constraint: corrupting them only shifts the transcript, which the FRI checks catch)
EXPECT = {
    "trace_cap": "init_challenger_state mismatch",
    "aux_cap": "Mismatch between evaluation and opening of quotient polynomial",
    "quotient_cap": "Mismatch between evaluation and opening of quotient polynomial",
    "local_values": "Mismatch between evaluation and opening of quotient polynomial",
    "auxiliary_polys": "Mismatch between evaluation and opening of quotient polynomial",
    "quotient_polys": "Mismatch between evaluation and opening of quotient polynomial",
    "init_state": "init_challenger_state mismatch",
}


@pytest.mark.parametrize("name", ["trace_cap", "aux_cap", "quotient_cap", "local_values", "next_values", "auxiliary_polys",
                                  "auxiliary_polys_next", "ctl_zs_first", "quotient_polys", "fri_caps", "queries",
                                  "final_poly", "pow_witness", "init_state"])
def test_rejects_corruption_of_every_section(gpu_ctx, oracle, g1, name):
    first, ln = sections(0, g1["words"].size)[name]
    rng = np.random.default_rng(hash(name) % (1 << 32))
    for _ in range(3):
        w = first + int(rng.integers(ln))
        bad = g1["words"].copy()
        bad[w] ^= np.uint64(1)
        mine, ref = both_verify(gpu_ctx, oracle, g1, words=bad)
        assert mine is not None, (name, w)
        assert ref is not None, (name, w)
        if name in EXPECT:
            assert mine == EXPECT[name], (name, w, mine)
        assert mine == ref, (name, w, mine, ref)  # same check trips in the same order as in the restated reference verifier


def test_rejects_wrong_public_values(gpu_ctx, oracle, g1):
    """The proof is fine but the claimed inputs/outputs are not the proven ones: CTL sums differ (check_ctls)."""
    s2 = g1["s"].copy()
    s2[5, 0] ^= np.uint64(2)
    mine, ref = both_verify(gpu_ctx, oracle, g1, s=s2)
    assert mine == "CTL sum mismatch" and ref == "CTL sum mismatch"
    out2 = g1["outputs"].copy()
    out2[17] ^= np.uint64(1)
    mine, _ = both_verify(gpu_ctx, oracle, g1, outputs=out2)
    assert mine == "CTL sum mismatch"
    x2 = g1["x"].copy()
    x2[0], x2[1] = g1["x"][1], g1["x"][0]
    mine, ref = both_verify(gpu_ctx, oracle, g1, x=x2)
    assert mine == "CTL sum mismatch" and ref == "CTL sum mismatch"


def test_rejects_wrong_shape_and_arguments(gpu_ctx, g1):
    with pytest.raises(pk.VerifyError, match="bad proof shape"):
        gpu_ctx.verify(0, g1["words"][:-1], g1["degree_bits"], g1["s"], g1["x"], g1["o"], g1["outputs"])
    with pytest.raises(pk.VerifyError, match="bad proof shape"):
        gpu_ctx.verify(0, g1["words"], g1["degree_bits"] + 1, g1["s"], g1["x"], g1["o"], g1["outputs"])
    with pytest.raises(pk.VerifyError, match="bad proof shape"):
        gpu_ctx.verify(2, g1["words"], g1["degree_bits"], g1["s"], g1["x"][:, :4].copy(), None, g1["outputs"][:128 * 4].copy())
    bad = g1["words"].copy()
    bad[300] = np.uint64(0xFFFFFFFF00000001)  # = p, not canonical
    with pytest.raises(pk.VerifyError, match="non-canonical"):
        gpu_ctx.verify(0, bad, g1["degree_bits"], g1["s"], g1["x"], g1["o"], g1["outputs"])
    with pytest.raises(RuntimeError, match="-1"):
        gpu_ctx.verify(7, g1["words"], g1["degree_bits"], g1["s"], g1["x"], g1["o"], g1["outputs"])


def test_padded_and_tall_g1(gpu_ctx, oracle):
    for n in (3, 200):  # 3 instances padded to 2^16 rows; 200 instances -> 2^17 rows, four FRI layers
        s, x, o = synth.g1_inputs(n, seed=99 + n)
        pr = gpu_ctx.prove_g1(s, x, o)
        p = dict(kind=0, s=s, x=x, o=o, words=pr.words, outputs=pr.outputs, degree_bits=pr.degree_bits)
        gpu_ctx.verify(0, pr.words, pr.degree_bits, s, x, o, pr.outputs)
        bad = pr.words.copy()
        bad[64 * 3 + 2 * 781 + 9] ^= np.uint64(4)
        with pytest.raises(pk.VerifyError, match="Mismatch between evaluation"):
            gpu_ctx.verify(0, bad, pr.degree_bits, s, x, o, pr.outputs)
        if n == 3:
            mine, ref = both_verify(gpu_ctx, oracle, p)
            assert mine is None and ref is None


@pytest.mark.parametrize("kind", [1, 2])
def test_g2_and_fq_exp(gpu_ctx, oracle, kind):
    if kind == 1:
        s, x, o = synth.g2_inputs(4)
        pr = gpu_ctx.prove_g2(s, x, o)
    else:
        s, x = synth.fq_inputs(6)
        o = None
        pr = gpu_ctx.prove_fq_exp(s, x)
    p = dict(kind=kind, s=s, x=x, o=o, words=pr.words.copy(), outputs=pr.outputs.copy(), degree_bits=pr.degree_bits)
    mine, ref = both_verify(gpu_ctx, oracle, p)
    assert mine is None and ref is None, (mine, ref)
    secs = sections(kind, p["words"].size)
    for name in ("local_values", "auxiliary_polys_next", "quotient_polys", "queries", "final_poly", "ctl_zs_first"):
        first, ln = secs[name]
        bad = p["words"].copy()
        bad[first + ln // 2] ^= np.uint64(1)
        mine, ref = both_verify(gpu_ctx, oracle, p, words=bad)
        assert mine is not None and mine == ref, (kind, name, mine, ref)
    s2 = s.copy()
    s2[1, 3] ^= np.uint64(1 << 40)
    mine, ref = both_verify(gpu_ctx, oracle, p, s=s2)
    assert mine == "CTL sum mismatch" and ref == "CTL sum mismatch"


def test_non_default_query_count_and_grinding(gpu_ctx):
    """num_queries and pow_bits of StarkConfig are parameters of both the prover and the verifier (FriConfig): a proof made with
    20 queries and 10 grinding bits verifies under the same parameters only."""
    s, x, o = synth.g1_inputs(3, seed=61)
    p = pk.default_params()
    p.num_queries, p.pow_bits = 20, 10
    pr = gpu_ctx.prove_g1(s, x, o, params=p)
    assert pr.words.size < 135841
    gpu_ctx.verify(0, pr.words, pr.degree_bits, s, x, o, pr.outputs, params=p)
    with pytest.raises(pk.VerifyError, match="bad proof shape"):
        gpu_ctx.verify(0, pr.words, pr.degree_bits, s, x, o, pr.outputs)
    q = pk.default_params()
    q.num_queries, q.pow_bits = 20, 24  # same shape, stricter grinding: the witness found for 10 bits does not pass
    with pytest.raises(pk.VerifyError, match="proof of work"):
        gpu_ctx.verify(0, pr.words, pr.degree_bits, s, x, o, pr.outputs, params=q)
