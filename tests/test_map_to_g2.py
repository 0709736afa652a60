"""map_to_g2 front-end (BASELINE.json configs[4]): the host arithmetic around the two STARK job kinds, and the pipeline
fq_exp proofs -> Legendre results -> G2 cofactor-clearing proofs on the GPU (src/utils/hash_to_g2.rs:113-148,150-207)."""
import numpy as np
import pytest

from tools import map_to_g2_ref as m2g
from tools import synth


def on_twist(pt):
    return synth.f2_mul(pt[1], pt[1]) == m2g.g(pt[0])


def test_square_roots_and_constants():
    rng = synth.Xoshiro256ss(7)
    n_sq = 0
    for _ in range(40):
        a = (rng.next_u256() % synth.P, rng.next_u256() % synth.P)
        r = m2g.f2_sqrt(a)
        is_sq = m2g.fq_is_square(m2g.f2_norm(a))  # a is a square in Fq2 iff its norm is a square in Fq
        assert (r is not None) == is_sq
        if r is not None:
            assert synth.f2_mul(r, r) == a
            n_sq += 1
    assert 5 < n_sq < 35
    assert m2g.f2_sqrt((4, 0)) in ((2, 0), (synth.P - 2, 0))
    r = m2g.f2_sqrt((synth.P - 4, 0))  # -4 = (2u)^2
    assert r is not None and synth.f2_mul(r, r) == (synth.P - 4, 0)
    assert synth.f2_mul(m2g._TV4, m2g._TV4) == synth.f2_mul(m2g.f2_neg(m2g._GZ), (3, 0))
    assert on_twist(synth.G2_GEN)


def test_candidates_and_cofactor_clearing():
    us = m2g.inputs(6, seed=11)
    fs, fx = m2g.fq_exp_jobs(us)
    assert fs.shape == (12, 4) and fx.shape == (12, 4)
    legendre = []
    for k in range(12):
        assert synth.words_to_int(fs[k]) == (synth.P - 1) // 2
        v = pow(synth.words_to_int(fx[k]), (synth.P - 1) // 2, synth.P)
        assert v in (1, synth.P - 1)
        legendre.append(v)
    gs, gx, goff, pts = m2g.g2_jobs(us, legendre, seed=3)
    for k, (pt, off) in enumerate(pts):
        assert on_twist(pt) and on_twist(off)
        assert m2g.sgn(pt[1]) == m2g.sgn(us[k])
        assert synth.words_to_int(gs[k]) == m2g.COFACTOR
        q = synth.g2_mul(m2g.COFACTOR, pt)  # cofactor * (point of the full twist group) lies in the r-torsion subgroup
        assert on_twist(q)
        assert synth.g2_mul(synth.R_ORDER - 1, q) == (q[0], m2g.f2_neg(q[1]))
    # at least one input takes each of the first two branches with overwhelming probability over 6 inputs
    assert any(legendre[2 * k] == 1 for k in range(6))


@pytest.mark.gpu
def test_pipeline_on_gpu(gpu_ctx):
    us = m2g.inputs(5, seed=23)
    fs, fx = m2g.fq_exp_jobs(us)
    pf = gpu_ctx.prove_fq_exp(fs, fx)
    legendre = [synth.words_to_int(w) for w in pf.outputs.reshape(-1, 4)]
    for k in range(10):
        assert legendre[k] == pow(synth.words_to_int(fx[k]), (synth.P - 1) // 2, synth.P)
    gpu_ctx.verify(2, pf.words, pf.degree_bits, fs, fx, None, pf.outputs)
    gs, gx, goff, pts = m2g.g2_jobs(us, legendre, seed=5)
    pg = gpu_ctx.prove_g2(gs, gx, goff)
    gpu_ctx.verify(1, pg.words, pg.degree_bits, gs, gx, goff, pg.outputs)
    outs = pg.outputs.reshape(-1, 16)
    for k, (pt, off) in enumerate(pts):
        assert m2g.finish(outs[k], off) == synth.g2_mul(m2g.COFACTOR, pt)


@pytest.mark.gpu
def test_device_front_end_matches_python(gpu_ctx):
    """bn254s_map_to_g2 (csrc/map_to_g2.hip): candidates, Legendre jobs, selected points with their signed square roots and
    the final images equal the Python front-end's, and every proof verifies against the job arrays Python derives."""
    import numpy as np
    n = 70  # 140 Legendre jobs: two Fq-exp proofs (128 + 12), one G2 proof
    us = m2g.inputs(n, seed=41)
    u = np.array([synth._to_words(a[0]) + synth._to_words(a[1]) for a in us], dtype=np.uint64)
    fs, fx = m2g.fq_exp_jobs(us)
    legendre = [pow(synth.words_to_int(w), (synth.P - 1) // 2, synth.P) for w in fx]
    gs, gx, goff, pts = m2g.g2_jobs(us, legendre, seed=9)
    out, fq_jobs, g2_jobs, pf, pg = gpu_ctx.map_to_g2(u, goff)
    assert np.array_equal(fq_jobs[:, :4], fs) and np.array_equal(fq_jobs[:, 4:], fx)
    assert np.array_equal(g2_jobs[:, :4], gs) and np.array_equal(g2_jobs[:, 4:], gx)
    assert len(pf) == 2 and len(pg) == 1
    picks = set()
    for k, (pt, off) in enumerate(pts):
        assert m2g.finish(pg[0].outputs.reshape(-1, 16)[k], off) == synth.g2_from_words(out[k])
        picks.add(0 if legendre[2 * k] == 1 else 1 if legendre[2 * k + 1] == 1 else 2)
    for k in range(0, n, 23):  # python G2 scalar multiplication is slow: spot-check the images
        assert synth.g2_from_words(out[k]) == synth.g2_mul(m2g.COFACTOR, pts[k][0])
    assert picks == {0, 1, 2}  # all three branches of the map taken
    for i, p in enumerate(pf):
        lo, hi = 128 * i, min(128 * (i + 1), 2 * n)
        gpu_ctx.verify(2, p.words, p.degree_bits, fs[lo:hi], fx[lo:hi], None, p.outputs)
    gpu_ctx.verify(1, pg[0].words, pg[0].degree_bits, gs, gx, goff, pg[0].outputs)


def test_hash_to_fq2_host_function_matches_python():
    """bn254s_hash_to_fq2 (host code of the library: Poseidon challenger + 512-bit reduction) against the pure-Python mirror,
    for inputs that end inside and on a rate boundary."""
    import numpy as np
    from plonky2_bn254_amd import lib as L
    lib = L.load_library()
    rng = synth.Xoshiro256ss(5)
    for n in (0, 1, 7, 8, 9, 16, 23):
        inp = np.array([rng.next_u64() % m2g.GL_P for _ in range(n)], dtype=np.uint64)
        out = np.zeros(8, np.uint64)
        assert lib.bn254s_hash_to_fq2(L._ptr(inp) if n else None, n, L._ptr(out)) == 0
        got = (synth.words_to_int(out[:4]), synth.words_to_int(out[4:]))
        assert got == m2g.hash_to_fq2(inp.tolist()), n
        assert got[0] < synth.P and got[1] < synth.P
    # extreme digits through the host permutation's sparse partial rounds (tools/derive_poseidon_host_fast.py)
    edge = [0, 1, m2g.GL_P - 1, (1 << 32) - 1, 1 << 32, m2g.GL_P - (1 << 32), 0xFFFFFFFF00000000]
    for k in range(len(edge)):
        inp = np.array([edge[(k + i) % len(edge)] for i in range(24)], dtype=np.uint64)
        out = np.zeros(8, np.uint64)
        assert lib.bn254s_hash_to_fq2(L._ptr(inp), inp.size, L._ptr(out)) == 0
        assert (synth.words_to_int(out[:4]), synth.words_to_int(out[4:])) == m2g.hash_to_fq2(inp.tolist()), k


@pytest.mark.gpu
def test_hash_to_fq2_batch_on_device_equals_host_and_python(gpu_ctx):
    """bn254s_hash_to_fq2_batch (one lane per message: device challenger + 512-bit reduction in the 26-bit-limb field) against the
    host function and the pure-Python mirror, for every input length class (empty, partial chunk, exactly one / two chunks,
    ragged), edge elements, and a batch of 4096 messages (the input count of configs[4])."""
    import plonky2_bn254_amd.lib as L
    lib = L.load_library()
    rng = np.random.default_rng(77)
    for ln in (0, 1, 7, 8, 9, 16, 21):
        inp = rng.integers(0, m2g.GL_P, size=(5, ln), dtype=np.uint64)
        if ln:
            inp[0, :] = m2g.GL_P - 1
            inp[1, :] = 0
        got = gpu_ctx.hash_to_fq2_batch(inp)
        for k in range(5):
            want = m2g.hash_to_fq2(inp[k].tolist())
            assert (synth.words_to_int(got[k, :4]), synth.words_to_int(got[k, 4:])) == want, (ln, k)
            host = np.zeros(8, np.uint64)
            row = np.ascontiguousarray(inp[k])
            assert lib.bn254s_hash_to_fq2(L._ptr(row) if ln else None, ln, L._ptr(host)) == 0
            assert np.array_equal(host, got[k])
    big = rng.integers(0, m2g.GL_P, size=(4096, 12), dtype=np.uint64)
    got = gpu_ctx.hash_to_fq2_batch(big)
    for k in (0, 1, 2047, 4095):
        assert (synth.words_to_int(got[k, :4]), synth.words_to_int(got[k, 4:])) == m2g.hash_to_fq2(big[k].tolist())
    assert len({r.tobytes() for r in got}) == 4096
    # and onward: the hashed values are valid inputs of bn254s_map_to_g2 (the reference's hash_to_g2 = the two together)
    _, _, off = synth.g2_inputs(4, seed=5)
    pts, fq_jobs, g2_jobs, pf, pg = gpu_ctx.map_to_g2(np.ascontiguousarray(got[:4]), off)
    gpu_ctx.verify(1, pg[0].words, pg[0].degree_bits, np.ascontiguousarray(g2_jobs[:, :4]), np.ascontiguousarray(g2_jobs[:, 4:]), off,
                   pg[0].outputs)
