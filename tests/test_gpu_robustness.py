"""Behaviour of the library around the happy path: errors leave the context usable, independent contexts may prove
concurrently from different host threads (the threading contract of include/bn254_stark.h), the doubling /
infinity cases of the parallel running-sum scan (csrc/chain_scan.h) agree with the sequential walk of the reference
(src/starks/curves/g1/scalar_mul_stark.rs:92-213) as restated by the oracle."""
import threading

import numpy as np
import pytest

import plonky2_bn254_amd as pk
from tools import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def neg_point_words(pt_words):
    y = synth.words_to_int(pt_words[4:])
    return list(pt_words[:4]) + [((synth.P - y) >> (64 * i)) & synth.MASK64 for i in range(4)]


def test_context_survives_an_invalid_point(gpu_ctx):
    s, x, o = synth.g1_inputs(4, seed=31)
    good = gpu_ctx.prove_g1(s, x, o).words.copy()
    bad_o = o.copy()
    bad_o[2] = neg_point_words(x[2])  # offset = -x: the very first addition meets the point at infinity
    with pytest.raises(RuntimeError, match="-4"):
        gpu_ctx.prove_g1(s, x, bad_o)
    with pytest.raises(RuntimeError, match="-4"):
        gpu_ctx.prove_g1_batch(np.tile(s, (40, 1)), np.tile(x, (40, 1)), np.tile(bad_o, (40, 1)))
    again = gpu_ctx.prove_g1(s, x, o)
    assert np.array_equal(again.words, good)
    gpu_ctx.verify(0, again.words, again.degree_bits, s, x, o, again.outputs)


def test_running_sum_meets_infinity_later_in_the_walk(gpu_ctx, oracle):
    """offset = -3x, bit 0 set: S_1 = -2x, so C_1 = S_1 + D_1 is the point at infinity at step 1 (whatever bit 1 is: the
    add row is generated either way): both the sequential walk and the scan must report it."""
    s, x, o = synth.g1_inputs(2, seed=77)
    xp = (synth.words_to_int(x[0, :4]), synth.words_to_int(x[0, 4:]))
    three_x = synth.g1_mul(3, xp)
    o[0] = synth._to_words(three_x[0]) + synth._to_words(synth.P - three_x[1])
    s[0] = [3, 0, 0, 0]
    with pytest.raises(RuntimeError, match="-4"):
        gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
    with pytest.raises(RuntimeError):
        oracle_lib.generate_trace(oracle, 0, s, x, o)
    # with bit 0 cleared S stays at -3x and never meets -D_k: fine for both, and cell-for-cell equal
    s[0] = [0xFFFFFFFFFFFFFFFE, 5, 0, 1 << 63]
    got, got_out = gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
    ref, ref_out = oracle_lib.generate_trace(oracle, 0, s, x, o)
    assert np.array_equal(got, ref) and np.array_equal(got_out.reshape(ref_out.shape), ref_out)


def test_scan_doubling_and_cancellation_cases(gpu_ctx, oracle):
    """Partial sums of the scan that double (x added to x) or cancel (a block of bits summing to a multiple of the group
    order) never occur in the sequential walk; the complete addition law must make them invisible in the trace."""
    s, x, o = synth.g1_inputs(6, seed=123)
    r = synth.R_ORDER
    s[0] = synth._to_words(r)             # s*x = infinity: result = offset, partial sums cancel on the way
    s[1] = synth._to_words(r + 1)
    s[2] = synth._to_words(2 * r)
    s[3] = synth._to_words((1 << 256) - 1)
    s[4] = synth._to_words(0)
    o[5] = x[5]                           # offset == x: the first addition is a doubling
    s[5] = synth._to_words(5)
    got, got_out = gpu_ctx.g1_generate_trace(s, x, o, min_rows_log2=16)
    ref, ref_out = oracle_lib.generate_trace(oracle, 0, s, x, o)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, bad[:5].tolist()
    assert np.array_equal(got_out.reshape(ref_out.shape), ref_out)
    for k in (0, 2, 4):
        assert np.array_equal(got_out.reshape(-1, 8)[k], o[k])


def test_two_contexts_prove_concurrently():
    s, x, o = synth.g1_inputs(128 * 3, seed=9)
    res = {}

    def work(tag):
        ctx = pk.Context(0)
        res[tag] = [p.words.copy() for p in ctx.prove_g1_batch(s, x, o)]
        ctx.close()

    th = [threading.Thread(target=work, args=(t,)) for t in ("a", "b")]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(res["a"]) == 3 and len(res["b"]) == 3
    for pa, pb in zip(res["a"], res["b"]):
        assert np.array_equal(pa, pb)


def test_plain_c_host_program(tmp_path):
    """examples/prove_g1.c: prove + verify + reject a corrupted proof through the C ABI alone."""
    import subprocess
    from tests.test_host_and_abi import build_c_example
    r = subprocess.run([build_c_example(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bn254s_verify: 0" in r.stdout and "corrupted proof: -8" in r.stdout


def test_batch_results_do_not_depend_on_the_number_of_slots(gpu_ctx, monkeypatch):
    """BN254S_SLOTS (proofs in flight) and the scheduler costs are throughput knobs only: the proofs must be the same."""
    s, x, o = synth.g1_inputs(128 * 2 + 5, seed=17)  # three proofs, the last one padded
    ref = None
    for slots, cap in (("1", None), ("3", "4"), ("8", None)):
        monkeypatch.setenv("BN254S_SLOTS", slots)
        proofs = gpu_ctx.prove_g1_batch(s, x, o)
        words = [p.words.copy() for p in proofs]
        assert len(words) == 3
        if ref is None:
            ref = words
        for a, b in zip(words, ref):
            assert np.array_equal(a, b), slots
    single = gpu_ctx.prove_g1(s[256:], x[256:], o[256:])
    assert np.array_equal(single.words, ref[2])


def test_multi_context_batch_matches_single_context(gpu_ctx):
    """bn254s_prove_batch_multi: proofs dealt round-robin to several contexts (one per GPU in production; two on the same GPU
    here) come back in order and identical to the one-context batch, including a short last proof."""
    s, x, o = synth.g1_inputs(128 * 4 + 9, seed=29)
    ref = [p.words.copy() for p in gpu_ctx.prove_g1_batch(s, x, o)]
    a, b = pk.Context(0), pk.Context(0)
    got = pk.prove_batch_multi([a, b], 0, s, x, o)
    assert len(got) == len(ref) == 5
    for g, r in zip(got, ref):
        assert np.array_equal(g.words, r)
    # errors of one context fail the call as a whole
    bad_o = o.copy()
    bad_o[130] = neg_point_words(x[130])
    with pytest.raises(RuntimeError, match="-4"):
        pk.prove_batch_multi([a, b], 0, s, x, bad_o)
    del got
    a.close()
    b.close()


def test_batches_in_flight_match_the_blocking_call(gpu_ctx):
    """bn254s_prove_batch_begin / _end: two batches of different kinds open at the same time come back in order and equal to the
    blocking call; a batch with a bad point fails in _end with the error code, the batch begun after it is unaffected and the
    context keeps working."""
    s, x, o = synth.g1_inputs(128 * 3 + 7, seed=41)
    fs, fx = synth.fq_inputs(128 + 3, seed=43)
    ref_g1 = [p.words.copy() for p in gpu_ctx.prove_g1_batch(s, x, o)]
    ref_fq = [p.words.copy() for p in gpu_ctx.prove_batch(2, fs, fx)]
    h1 = gpu_ctx.prove_batch_begin(0, s, x, o)
    h2 = gpu_ctx.prove_batch_begin(2, fs, fx)
    h3 = gpu_ctx.prove_batch_begin(0, s[:130], x[:130], o[:130])
    got_g1, got_fq, got_3 = h1.end(), h2.end(), h3.end()
    assert [len(got_g1), len(got_fq), len(got_3)] == [4, 2, 2]
    for g, r in zip(got_g1, ref_g1):
        assert np.array_equal(g.words, r)
    for g, r in zip(got_fq, ref_fq):
        assert np.array_equal(g.words, r)
    assert np.array_equal(got_3[0].words, ref_g1[0])
    with pytest.raises(RuntimeError, match="already ended"):
        h1.end()
    bad_o = o.copy()
    bad_o[130] = neg_point_words(x[130])
    hb = gpu_ctx.prove_batch_begin(0, s, x, bad_o)
    hg = gpu_ctx.prove_batch_begin(0, s[:128], x[:128], o[:128])
    with pytest.raises(RuntimeError, match="-4"):
        hb.end()
    good = hg.end()
    assert np.array_equal(good[0].words, ref_g1[0])
    assert np.array_equal(gpu_ctx.prove_g1(s[:128], x[:128], o[:128]).words, ref_g1[0])


def test_single_proof_calls_while_a_batch_is_open(gpu_ctx):
    """The threading rule of include/bn254_stark.h: between bn254s_prove_batch_begin and _end the context may be entered again;
    bn254s_prove_g1 / _fq_exp / bn254s_verify called meanwhile give the right proof (they queue on the same worker pool and
    never share a slot - stream, workspace, staging buffer - with a proof of the open batch), and the batch is unharmed."""
    s, x, o = synth.g1_inputs(128 * 6, seed=53)
    fs, fx = synth.fq_inputs(9, seed=54)
    ref_batch = [p.words.copy() for p in gpu_ctx.prove_g1_batch(s, x, o)]
    ref_single = gpu_ctx.prove_g1(s[:5], x[:5], o[:5]).words.copy()
    ref_fq = gpu_ctx.prove_fq_exp(fs, fx).words.copy()
    for rounds in range(2):
        h = gpu_ctx.prove_batch_begin(0, s, x, o)           # six proofs running on slots 0..5
        a = gpu_ctx.prove_g1(s[:5], x[:5], o[:5])           # a different shape (padded proof) in between
        b = gpu_ctx.prove_fq_exp(fs, fx)                    # and a different kind
        gpu_ctx.verify(0, a.words, 16, s[:5], x[:5], o[:5], a.outputs)
        h2 = gpu_ctx.prove_batch_begin(0, s[:256], x[:256], o[:256])
        gpu_ctx.trim()                                      # idle slots give their workspaces back; running proofs are left alone
        c = gpu_ctx.prove_g1(s[128:256], x[128:256], o[128:256])
        got, got2 = h.end(), h2.end()
        assert np.array_equal(a.words, ref_single) and np.array_equal(b.words, ref_fq)
        assert np.array_equal(c.words, ref_batch[1])
        assert len(got) == 6 and all(np.array_equal(g.words, r) for g, r in zip(got, ref_batch))
        assert all(np.array_equal(g.words, r) for g, r in zip(got2, ref_batch[:2]))


def test_closed_proof_keeps_its_words(gpu_ctx):
    s, x, o = synth.g1_inputs(2, seed=57)
    p = gpu_ctx.prove_g1(s, x, o)
    q = gpu_ctx.prove_g1(s, x, o)
    w = p.words.copy()
    q.close()                          # words / outputs are copied out before the library's copy is freed
    assert np.array_equal(q.words, w) and q.outputs.size == 16 and np.array_equal(q.caps(), w[:192])
    with pytest.raises(RuntimeError):
        q.section("trace_cap")
