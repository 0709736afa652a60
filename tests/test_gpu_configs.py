"""BASELINE.json configs at full size on the GPU, through the C ABI, with size-independent properties:
configs[1] 1024 G1 scalar-muls = 8 proofs (the steady state bench.py times), configs[2] 1024 G2 scalar-muls,
configs[4]'s sharding over several contexts.  Every proof is checked by the library's verifier (bn254s_verify) and
by the oracle's restatement of the reference's verify() (common/verifier.rs:32-98); outputs by Python big integers."""
import numpy as np
import pytest

import bench
import plonky2_bn254_amd as pk
from tools import synth
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def _w2p(w):
    return (synth.words_to_int(w[:4]), synth.words_to_int(w[4:]))


def test_config2_1024_g1_every_proof_verified(gpu_ctx, oracle):
    """hook.rs:63-71 batches all calls of a circuit; here 1024 calls as 8 x 128 (the reference's test shape,
    scalar_mul_stark.rs:554,569), inputs drawn exactly like bench.py draws a step."""
    _, xs, offs = synth.g1_inputs(1024, seed=bench.SEED0 + 1)
    s, x, o = bench.step_inputs((xs, offs), 3, 0)
    proofs = gpu_ctx.prove_g1_batch(s, x, o)
    assert len(proofs) == 8
    caps = bench.caps_of(proofs)
    assert caps.shape == (8, 192) and len({c.tobytes() for c in caps}) == 8      # eight different proofs
    for j, p in enumerate(proofs):
        sl = slice(128 * j, 128 * j + 128)
        gpu_ctx.verify(0, p.words, p.degree_bits, s[sl], x[sl], o[sl], p.outputs)
        rc, msg = oracle_lib.g1_verify(oracle, p.words, p.degree_bits, s[sl], x[sl], o[sl])
        assert rc == 0, (j, msg)
        for i in (0, 127):
            want = synth.g1_scalar_mul_offset(synth.words_to_int(s[sl][i]), _w2p(x[sl][i]), _w2p(o[sl][i]))
            assert _w2p(p.outputs.reshape(128, 8)[i]) == want
    # the batch path and the single-proof path give the same words; a proof does not verify against another proof's inputs
    single = gpu_ctx.prove_g1(s[640:768], x[640:768], o[640:768])
    assert np.array_equal(single.words, proofs[5].words)
    with pytest.raises(pk.VerifyError):
        gpu_ctx.verify(0, proofs[5].words, 16, s[:128], x[:128], o[:128], proofs[5].outputs)


def test_config3_1024_g2_every_proof_verified(gpu_ctx, oracle):
    """1024 G2 scalar multiplications (g2/scalar_mul_stark.rs): 128 synthetic base points / offsets (python big-int G2 is slow),
    1024 distinct random 256-bit scalars and a different point assignment per proof."""
    _, x0, o0 = synth.g2_inputs(128)
    rng = np.random.default_rng(20260)
    s = rng.integers(0, 1 << 63, size=(1024, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(1024, 4), dtype=np.uint64)
    x = np.concatenate([x0[rng.permutation(128)] for _ in range(8)])
    o = np.concatenate([o0[rng.permutation(128)] for _ in range(8)])
    proofs = gpu_ctx.prove_batch(1, s, x, o)
    assert len(proofs) == 8 and len({p.words[:192].tobytes() for p in proofs}) == 8
    for j, p in enumerate(proofs):
        sl = slice(128 * j, 128 * j + 128)
        gpu_ctx.verify(1, p.words, p.degree_bits, s[sl], x[sl], o[sl], p.outputs)
    rc, msg = oracle_lib.verify(oracle, 1, proofs[7].words, 16, s[896:], x[896:], o[896:])
    assert rc == 0, msg
    i = 5
    want = synth.g2_scalar_mul_offset(synth.words_to_int(s[i]), synth.g2_from_words(x[i]), synth.g2_from_words(o[i]))
    assert synth.g2_from_words(proofs[0].outputs.reshape(128, 16)[i]) == want
    single = gpu_ctx.prove_g2(s[256:384], x[256:384], o[256:384])
    assert np.array_equal(single.words, proofs[2].words)


def test_config5_jobs_shard_over_contexts(gpu_ctx):
    """configs[4] on several GPUs: the Fq-exp and G2 jobs of map_to_g2 (hash_to_g2.rs:184-201) dealt to two contexts by
    bn254s_prove_batch_multi (one context per GPU in production; two on this GPU) give the proofs bn254s_map_to_g2 makes on
    one context, in order; and a rank's contiguous share of the inputs (bench.split_range) gives that share of the outputs."""
    n = 160
    u, off = bench.map_to_g2_inputs(0, n)
    pts, fq_jobs, g2_jobs, pf, pg = gpu_ctx.map_to_g2(u, off)
    assert (len(pf), len(pg)) == bench.map_to_g2_proof_counts(n) == (3, 2)
    a, b = pk.Context(0), pk.Context(0)
    fs, fx = np.ascontiguousarray(fq_jobs[:, :4]), np.ascontiguousarray(fq_jobs[:, 4:])
    gs, gx = np.ascontiguousarray(g2_jobs[:, :4]), np.ascontiguousarray(g2_jobs[:, 4:])
    got_f = pk.prove_batch_multi([a, b], 2, fs, fx)
    got_g = pk.prove_batch_multi([a, b], 1, gs, gx, off)
    for g, r in zip(got_f + got_g, pf + pg):
        assert np.array_equal(g.words, r.words) and np.array_equal(g.outputs, r.outputs)
    # rank 1 of 2: its slice alone reproduces its part of the result (no cross-rank state)
    lo, hi = bench.split_range(1, 2, n)
    u1, off1 = bench.map_to_g2_inputs(lo, hi)
    assert np.array_equal(u1, u[lo:hi]) and np.array_equal(off1, off[lo:hi])
    pts1, fq1, g21, pf1, pg1 = a.map_to_g2(u1, off1)
    assert np.array_equal(pts1, pts[lo:hi]) and np.array_equal(fq1, fq_jobs[2 * lo:2 * hi]) and np.array_equal(g21, g2_jobs[lo:hi])
    for i, p in enumerate(pf1):
        sl = slice(128 * i, 128 * i + 128)
        a.verify(2, p.words, p.degree_bits, np.ascontiguousarray(fq1[sl, :4]), np.ascontiguousarray(fq1[sl, 4:]), None, p.outputs)
    del got_f, got_g, pf1, pg1
    a.close()
    b.close()
