"""GPU parity for tall traces (more than 128 calls in ONE proof, as `Bn254Hook::constrain` produces for a circuit with many
calls: reference src/hook.rs:63-71, rows = (512 n).next_power_of_two(), scalar_mul_stark.rs:60): N = 2^17 and 2^18."""
import numpy as np
import pytest

from tools import synth
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


def test_split_level_2pow18_matches_oracle(gpu_ctx, oracle, monkeypatch):
    """The 2^23-row path (one radix-2 level above the tall transforms, chunked commitment, in-place auxiliary commitment,
    halves opened at zeta^2) forced at 2^18 rows, where the oracle can still prove: every word of the proof."""
    s, x = synth.fq_inputs(300, seed=33)
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 2, s, x)
    monkeypatch.setenv("BN254S_FORCE_SPLIT", "1")
    pr = gpu_ctx.prove_fq_exp(s, x)
    monkeypatch.delenv("BN254S_FORCE_SPLIT")
    assert pr.degree_bits == 18 and pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing words {bad[:5]} of {ref.size}"
    assert np.array_equal(pr.outputs.reshape(-1, 4), ref_out)


def test_compact_workspace_2pow17_matches_oracle(gpu_ctx, oracle, monkeypatch):
    """The compact workspace of the proofs that would not fit otherwise (G1 at 2^23 rows, G2 from 2^22 rows on: NTT stage in
    column chunks, auxiliary commitment in place, auxiliary LDE in the memory of the trace values) forced at 2^17 rows."""
    s, x, o = synth.g1_inputs(150, seed=34)
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, 0, s, x, o)
    monkeypatch.setenv("BN254S_FORCE_LOWMEM", "1")
    pr = gpu_ctx.prove_g1(s, x, o)
    monkeypatch.delenv("BN254S_FORCE_LOWMEM")
    assert degree_bits == 17 and pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing words {bad[:5]} of {ref.size}"
    pr2 = gpu_ctx.prove_g1(s, x, o)          # back to the plain layout in the same slot
    assert np.array_equal(pr2.words, ref)


@pytest.mark.parametrize("kind,win_log,split", [(0, 15, False), (1, 16, False), (2, 16, True)])
def test_streaming_workspace_matches_oracle(gpu_ctx, oracle, monkeypatch, kind, win_log, split):
    """The workspace of the proofs whose LDEs do not fit the device (G2 at 2^23 rows): only coefficients resident, the commitment
    streamed chunk by chunk through the NTT into a leaf hash that absorbs into resident sponge states, the quotient over windows of
    consecutive rows recomputed from the coefficients (natural order, transition row from the next window), the FRI batch
    polynomial formed on the coefficient vectors, query rows from one more pass of chunk LDEs - forced at 2^17 rows (G1: four
    windows per coset; G2: two) and, with the radix-2 split level on top, at 2^18 (Fq-exp): every word of the proof."""
    if kind == 0:
        ins = synth.g1_inputs(150, seed=35)
    elif kind == 1:
        ins = synth.g2_inputs(130, seed=36)
    else:
        ins = synth.fq_inputs(300, seed=37)
    off = ins[2] if len(ins) > 2 else None
    ref, ref_out, _, degree_bits = oracle_lib.prove(oracle, kind, ins[0], ins[1], off)
    monkeypatch.setenv("BN254S_FORCE_STREAM", "1")
    monkeypatch.setenv("BN254S_STREAM_WIN_LOG", str(win_log))
    if split:
        monkeypatch.setenv("BN254S_FORCE_SPLIT", "1")
    pr = gpu_ctx.prove_batch(kind, ins[0], ins[1], off, per_proof=ins[0].shape[0])[0]
    monkeypatch.delenv("BN254S_FORCE_STREAM")
    monkeypatch.delenv("BN254S_STREAM_WIN_LOG")
    if split:
        monkeypatch.delenv("BN254S_FORCE_SPLIT")
    assert pr.degree_bits == degree_bits == (18 if split else 17) and pr.words.shape == ref.shape
    bad = np.flatnonzero(pr.words != ref)
    assert bad.size == 0, f"first differing words {bad[:5]} of {ref.size}"
    assert np.array_equal(pr.outputs.reshape(ref_out.shape), ref_out)
    pr2 = gpu_ctx.prove_batch(kind, ins[0], ins[1], off, per_proof=ins[0].shape[0])[0]     # back to the resident layout, same slot
    assert np.array_equal(pr2.words, ref)


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


def test_fq_exp_very_tall_proofs_are_accepted_by_both_verifiers(gpu_ctx, oracle):
    """N = 2^21 (R = 32 blocks per column: the outer DFT uses w_32 = 2^6) and N = 2^20: too tall for the CPU oracle to
    prove within a test, so the proof is checked by the oracle's verifier (independent AIR restatement at zeta, FRI, CTL
    sums) and by the library's own; a corrupted opening is rejected by both."""
    for n, bits in ((4096, 21), (1100, 20)):
        s, x = synth.fq_inputs(n, seed=40 + bits)
        pr = gpu_ctx.prove_fq_exp(s, x)
        assert pr.degree_bits == bits
        for k in range(0, n, 397):
            assert synth.words_to_int(pr.outputs.reshape(-1, 4)[k]) == pow(synth.words_to_int(x[k]), synth.words_to_int(s[k]), synth.P)
        rc, msg = oracle_lib.verify(oracle, 2, pr.words, bits, s, x)
        assert rc == 0, msg
        gpu_ctx.verify(2, pr.words, bits, s, x, None, pr.outputs)
        bad = pr.words.copy()
        bad[64 * 3 + 2 * 427 + 2 * 427 + 5] ^= np.uint64(1)  # an auxiliary opening
        rc, msg = oracle_lib.verify(oracle, 2, bad, bits, s, x)
        assert rc == 1 and "Mismatch" in msg


def test_fq_exp_2pow23_rows_one_proof(gpu_ctx):
    """16384 calls in ONE proof (N = 2^23, the size Bn254Hook::constrain would produce for BASELINE's 16384-call batch as a single
    circuit): the radix-2 level above the tall transforms at its real size.  Checked by both of the library's verifiers (GPU
    constraint sum and the independent host statement of the AIR); a corrupted opening is rejected."""
    n = 16384
    s, x = synth.fq_inputs(n, seed=63)
    pr = gpu_ctx.prove_fq_exp(s, x)
    assert pr.degree_bits == 23
    for k in range(0, n, 1543):
        assert synth.words_to_int(pr.outputs.reshape(-1, 4)[k]) == pow(synth.words_to_int(x[k]), synth.words_to_int(s[k]), synth.P)
    import plonky2_bn254_amd as pk
    gpu_ctx.verify(2, pr.words, 23, s, x, None, pr.outputs)
    pk.verify_host(2, pr.words, 23, s, x, None, pr.outputs)
    bad = pr.words.copy()
    bad[64 * 3 + 2 * 427 + 2 * 427 + 5] ^= np.uint64(1)
    with pytest.raises(pk.VerifyError):
        gpu_ctx.verify(2, bad, 23, s, x, None, pr.outputs)
    with pytest.raises(pk.VerifyError):
        pk.verify_host(2, bad, 23, s, x, None, pr.outputs)


def test_g2_2pow23_rows_one_proof(gpu_ctx):
    """16384 G2 scalar multiplications in ONE proof (hook.rs:63-71 with g2/scalar_mul_stark.rs:60): its two LDEs alone are 296 GB, so
    the proof runs in the streaming workspace (coefficients resident: 148 GB; quotient in four windows of 2^22 rows).  128 distinct
    points tiled, 16384 distinct scalars; checked by both of the library's verifiers, outputs by Python big integers, a corrupted
    opening rejected."""
    import plonky2_bn254_amd as pk
    n = 16384
    _, x0, o0 = synth.g2_inputs(128, seed=71)
    rng = np.random.default_rng(2323)
    s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x = np.ascontiguousarray(x0[rng.integers(0, 128, size=n)])
    o = np.ascontiguousarray(o0[rng.integers(0, 128, size=n)])
    gpu_ctx.trim()
    pr = gpu_ctx.prove_g2(s, x, o)
    assert pr.degree_bits == 23
    print("G2 2^23 rows: stage ms", {k: round(v, 1) for k, v in pr.stage_ms.items()})
    for i in (0, 9999, 16383):
        want = synth.g2_scalar_mul_offset(synth.words_to_int(s[i]), synth.g2_from_words(x[i]), synth.g2_from_words(o[i]))
        assert synth.g2_from_words(pr.outputs.reshape(n, 16)[i]) == want
    gpu_ctx.verify(1, pr.words, 23, s, x, o, pr.outputs)
    pk.verify_host(1, pr.words, 23, s, x, o, pr.outputs)
    bad = pr.words.copy()
    bad[64 * 3 + 2 * 1295 + 2 * 1295 + 5] ^= np.uint64(1)  # an auxiliary opening
    with pytest.raises(pk.VerifyError):
        pk.verify_host(1, bad, 23, s, x, o, pr.outputs)
    del pr
    gpu_ctx.trim()
    s2, x2 = synth.fq_inputs(3, seed=72)
    p2 = gpu_ctx.prove_fq_exp(s2, x2)
    gpu_ctx.verify(2, p2.words, p2.degree_bits, s2, x2, None, p2.outputs)


def test_out_of_memory_is_an_error_and_the_context_survives(gpu_ctx, monkeypatch):
    """A proof that finds no device memory returns BN254S_E_OOM (-3), its partial workspace is released and the context goes on
    proving.  (A memory reserve larger than the device makes every workspace "too large": the check that keeps room for the
    runtime's own allocations, prover.hip.)"""
    import plonky2_bn254_amd as pk
    s, x, o = synth.g1_inputs(3, seed=73)
    ctx = pk.Context(0)
    good = ctx.prove_g1(s, x, o).words.copy()
    ctx.close()
    # (the reserve is read once per process: a child process shows the error path through the C ABI)
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import plonky2_bn254_amd as pk
        from tools import synth
        s, x, o = synth.g1_inputs(3, seed=73)
        ctx = pk.Context(0)
        try:
            ctx.prove_g1(s, x, o)
            print("NO ERROR")
        except RuntimeError as e:
            print("ERR", "-3" in str(e), "reserve" in str(e))
        try:
            ctx.prove_batch(0, s, x, o)
            print("NO ERROR")
        except RuntimeError as e:
            print("ERR", "-3" in str(e))
    """ % str(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
    env = dict(__import__("os").environ, BN254S_MEM_RESERVE_MB="400000")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.stdout.split("\n")[:2] == ["ERR True True", "ERR True"], r.stdout + r.stderr
    s2, x2 = synth.fq_inputs(3, seed=72)
    pr = gpu_ctx.prove_fq_exp(s2, x2)
    gpu_ctx.verify(2, pr.words, pr.degree_bits, s2, x2, None, pr.outputs)
    assert good.size > 0


def test_batch_of_tall_proofs_larger_than_memory_runs_as_many_at_a_time_as_fit(gpu_ctx):
    """Ten G2 proofs of 1024 instances (2^19 rows, ~35 GB of workspace each) in one batch: more than the device holds with every
    slot busy, so proofs that find no memory give back idle workspaces, wait for a running proof to finish and try again; the
    batch completes and its proofs verify."""
    base = synth.g2_inputs(64, seed=81)
    s, x, o = (np.tile(a, (160, 1)) for a in base)      # 10240 instances
    proofs = gpu_ctx.prove_batch(1, s, x, o, per_proof=1024)
    assert len(proofs) == 10 and all(p.degree_bits == 19 for p in proofs)
    for j in (0, 9):
        sl = slice(1024 * j, 1024 * (j + 1))
        gpu_ctx.verify(1, proofs[j].words, 19, s[sl], x[sl], o[sl], proofs[j].outputs)
    # the context goes back to small proofs afterwards
    s2, x2 = synth.fq_inputs(3, seed=82)
    pr = gpu_ctx.prove_batch(2, s2, x2)[0]
    gpu_ctx.verify(2, pr.words, pr.degree_bits, s2, x2, None, pr.outputs)
