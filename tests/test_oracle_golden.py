"""CPU tests: the oracle against golden vectors and algebraic identities (no GPU needed)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tools import synth
from tests import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = 2**64 - 2**32 + 1
Q = synth.P


def words(v, n=4):
    return np.array([(v >> (64 * i)) & synth.MASK64 for i in range(n)], dtype=np.uint64)


def test_poseidon_kats(oracle):
    kat = json.load(open(os.path.join(GOLD, "poseidon_kat.json")))
    inputs = {"zeros": [0] * 12, "range12": list(range(12)), "neg_ones": [P - 1] * 12}
    for k in kat["kats"]:
        st = np.array(inputs[k["input"]], dtype=np.uint64)
        oracle.orc_poseidon_permute(oracle_lib.ptr(st))
        assert " ".join("%016x" % int(v) for v in st) == k["output"]
    for fn in ("oracle/poseidon_constants.inc", "plonky2_bn254_amd/csrc/poseidon_constants.inc"):
        txt = open(os.path.join(os.path.dirname(GOLD), "..", fn)).read()
        consts = [int(t.rstrip("ULL,"), 16) for t in txt.split() if t.startswith("0x")]
        assert len(consts) == 360
        digest = hashlib.sha256(b"".join(int(c).to_bytes(8, "little") for c in consts)).hexdigest()
        assert digest == kat["round_constants_sha256_le_u64"], fn


def test_sparse_partial_round_permutation_equals_the_textbook_one(oracle):
    """The oracle hashes with poseidon_permute_fast (oracle/hash.hpp: the 22 partial rounds in sparse form); the restatement
    proper is the textbook poseidon_permute, pinned by the upstream KATs above.  Same outputs on the KATs, on states built
    from the edge values of every reduction (0, 1, p-1, 2^32-1, 2^32, p-2^32) and on random states; and a whole commitment
    (from_values: NTT, leaf hashing, tree) is the same words with either."""
    import random
    kat = json.load(open(os.path.join(GOLD, "poseidon_kat.json")))
    inputs = {"zeros": [0] * 12, "range12": list(range(12)), "neg_ones": [P - 1] * 12}
    for k in kat["kats"]:
        st = np.array(inputs[k["input"]], dtype=np.uint64)
        oracle.orc_poseidon_permute_fast(oracle_lib.ptr(st))
        assert " ".join("%016x" % int(v) for v in st) == k["output"]
    rng = random.Random(11)
    edge = [0, 1, P - 1, (1 << 32) - 1, 1 << 32, P - (1 << 32), P - 2, (1 << 63)]
    for _ in range(4000):
        vals = [rng.choice(edge) if rng.random() < 0.5 else rng.randrange(P) for _ in range(12)]
        a = np.array(vals, dtype=np.uint64)
        b = a.copy()
        oracle.orc_poseidon_permute(oracle_lib.ptr(a))
        oracle.orc_poseidon_permute_fast(oracle_lib.ptr(b))
        assert np.array_equal(a, b), vals
    vals = np.random.default_rng(5).integers(0, 2**63, size=(9, 65536), dtype=np.uint64)
    fast = oracle_lib.commit_values(oracle, vals)
    oracle.orc_set_fast_poseidon(0)
    try:
        plain = oracle_lib.commit_values(oracle, vals)
    finally:
        oracle.orc_set_fast_poseidon(1)
    assert all(np.array_equal(f, p) for f, p in zip(fast, plain))


def test_goldilocks_field_ops(oracle):
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, b = int(rng.integers(0, 2**63)) * 2 % P, int(rng.integers(0, 2**63)) % P
        assert oracle.orc_gl_mul(a, b) == a * b % P
    edge = [0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - (1 << 32), 0xFFFFFFFF00000000, 1 << 63]
    for a in edge:          # every carry / borrow / ">= p" case of the 128-bit reduction
        for b in edge:
            assert oracle.orc_gl_mul(a, b) == a * b % P, (a, b)
    for a in (1, 2, P - 1, 0xc65c18b67785d900, 12345678901234567):
        assert oracle.orc_gl_inv(a) == pow(a, -1, P)


def test_bn254_golden_vectors(oracle):
    vec = json.load(open(os.path.join(GOLD, "bn254_kat.json")))
    out = np.zeros(4, np.uint64)
    for a, b, c in vec["fq_mul"]:
        oracle.orc_fq_mul(oracle_lib.ptr(words(int(a))), oracle_lib.ptr(words(int(b))), oracle_lib.ptr(out))
        assert synth.words_to_int(out) == int(c)
    for a, ai in vec["fq_inv"]:
        oracle.orc_fq_inv(oracle_lib.ptr(words(int(a))), oracle_lib.ptr(out))
        assert synth.words_to_int(out) == int(ai)
    out8 = np.zeros(8, np.uint64)
    for a, b, c in vec["g1_add"]:
        pa = np.concatenate([words(int(a[0])), words(int(a[1]))])
        pb = np.concatenate([words(int(b[0])), words(int(b[1]))])
        assert oracle.orc_g1_add(oracle_lib.ptr(pa), oracle_lib.ptr(pb), oracle_lib.ptr(out8)) == 0
        assert (synth.words_to_int(out8[:4]), synth.words_to_int(out8[4:])) == (int(c[0]), int(c[1]))


@pytest.mark.parametrize("n", [1, 2, 8, 64])
def test_ntt_against_naive_dft(oracle, n):
    rng = np.random.default_rng(n)
    c = [int(v) % P for v in rng.integers(0, 2**63, size=n)]
    w = pow(0x64fdd1a46201e246, 2 ** (32 - (n.bit_length() - 1)), P)   # primitive n-th root
    want = [sum(c[i] * pow(w, i * j, P) for i in range(n)) % P for j in range(n)]
    a = np.array(c, dtype=np.uint64)
    oracle.orc_ntt(oracle_lib.ptr(a), n, 0, 0)
    assert [int(v) for v in a] == want
    oracle.orc_ntt(oracle_lib.ptr(a), n, 1, 0)          # ifft(fft(c)) == c
    assert [int(v) for v in a] == c
    shift = 0xc65c18b67785d900
    b = np.array(c, dtype=np.uint64)
    oracle.orc_ntt(oracle_lib.ptr(b), n, 2, shift)      # coset fft = evaluation at shift * w^j
    assert [int(v) for v in b] == [sum(c[i] * pow(shift * pow(w, j, P), i, P) for i in range(n)) % P for j in range(n)]
    oracle.orc_ntt(oracle_lib.ptr(b), n, 3, shift)
    assert [int(v) for v in b] == c


def test_merkle_cap_and_hash_or_noop(oracle):
    # leaves of <= 4 elements are not hashed (hash_or_noop); cap of a 32-leaf tree = 16 two_to_one nodes
    leaves = np.arange(32 * 3, dtype=np.uint64).reshape(32, 3)
    cap = np.zeros((16, 4), np.uint64)
    oracle.orc_merkle_cap(oracle_lib.ptr(leaves), 32, 3, 4, oracle_lib.ptr(cap))
    for i in range(16):
        l = np.array(list(leaves[2 * i]) + [0], dtype=np.uint64)
        r = np.array(list(leaves[2 * i + 1]) + [0], dtype=np.uint64)
        st = np.concatenate([l, r, np.zeros(4, np.uint64)])
        oracle.orc_poseidon_permute(oracle_lib.ptr(st))
        assert np.array_equal(cap[i], st[:4])
    # hash_no_pad: overwrite-mode sponge over 8-element chunks, last chunk keeps the old rate lanes
    x = np.arange(1, 12, dtype=np.uint64)
    d = np.zeros(4, np.uint64)
    oracle.orc_hash_or_noop(oracle_lib.ptr(x), 11, oracle_lib.ptr(d))
    st = np.zeros(12, np.uint64)
    st[:8] = x[:8]
    oracle.orc_poseidon_permute(oracle_lib.ptr(st))
    st[:3] = x[8:]
    oracle.orc_poseidon_permute(oracle_lib.ptr(st))
    assert np.array_equal(d, st[:4])


def test_challenger_pop_order(oracle):
    # 8 observations trigger one duplexing; challenges pop from lane 7 downwards
    obs = np.arange(10, 18, dtype=np.uint64)
    out = np.zeros(3, np.uint64)
    oracle.orc_challenger_observe_get(oracle_lib.ptr(obs), 8, oracle_lib.ptr(out), 3)
    st = np.zeros(12, np.uint64)
    st[:8] = obs
    oracle.orc_poseidon_permute(oracle_lib.ptr(st))
    assert [int(v) for v in out] == [int(st[7]), int(st[6]), int(st[5])]


def test_modulus_zero_identity(oracle):
    """generate_modulus_zero output satisfies assert_modulus_zero (modulus_zero.rs:126-160) over the integers."""
    rng = synth.Xoshiro256ss(5)
    mod_limbs = [(Q >> (16 * i)) & 0xFFFF for i in range(16)]
    for case in range(10):
        a, b = rng.next_u256() % Q, rng.next_u256() % Q
        c = a * b % Q
        al = [(a >> (16 * i)) & 0xFFFF for i in range(16)]
        bl = [(b >> (16 * i)) & 0xFFFF for i in range(16)]
        cl = [(c >> (16 * i)) & 0xFFFF for i in range(16)]
        diff = [0] * 31
        for i in range(16):
            for j in range(16):
                diff[i + j] += al[i] * bl[j]
        for i in range(16):
            diff[i] -= cl[i]
        if case % 2:   # negative quotient as well
            diff = [-d for d in diff]
        inp = np.array(diff, dtype=np.int64)
        out = np.zeros(80, np.uint64)
        assert oracle.orc_generate_modulus_zero(oracle_lib.ptr(inp), oracle_lib.ptr(out)) == 0
        iqp, quot_abs = int(out[0]), [int(v) for v in out[1:18]]
        lo, hi = [int(v) for v in out[18:49]], [int(v) for v in out[49:80]]
        assert all(v < 65536 for v in quot_abs + lo + hi)
        sign = 2 * iqp - 1
        constr = [0] * 32
        for i in range(17):
            for j in range(16):
                constr[i + j] += sign * quot_abs[i] * mod_limbs[j]
        aux = [lo[i] - (1 << 29) + (hi[i] << 16) for i in range(31)] + [0]
        constr[0] += -(1 << 16) * aux[0]
        for d in range(1, 32):
            constr[d] += aux[d - 1] - (1 << 16) * aux[d]
        for i in range(31):
            constr[i] -= diff[i]
        assert constr == [0] * 32
        total = sum(d << (16 * i) for i, d in enumerate(diff))
        assert total % Q == 0 and (iqp == 1) == (total > 0)
    bad = np.zeros(31, np.int64)
    bad[0] = 1   # not a multiple of p: the reference asserts (modulus_zero.rs:82)
    assert oracle.orc_generate_modulus_zero(oracle_lib.ptr(bad), oracle_lib.ptr(np.zeros(80, np.uint64))) == -1


def test_g1_trace_satisfies_air_and_layout(oracle):
    """Every transition of a generated trace satisfies the 1111 constraints; column map as in
    scalar_mul_view.rs (row_position_correctness: INPUT_FILTER 770, OUTPUT_FILTER 771, TIMESTAMP 775,
    FREQ 779, RANGE_COUNTER 780)."""
    assert oracle.orc_g1_width() == 781
    s, x, o = synth.g1_inputs(2, seed=3)
    o[1] = x[1]   # doubling branch on an addition row
    rows = oracle.orc_g1_num_rows(2, 16)
    tr = np.zeros((781, rows), np.uint64)
    outs = np.zeros((2, 8), np.uint64)
    assert oracle.orc_g1_generate_trace(oracle_lib.ptr(s), oracle_lib.ptr(x), oracle_lib.ptr(o), 2, 16, oracle_lib.ptr(tr),
                                        oracle_lib.ptr(outs)) == 0
    assert tr[770, 0] == 1 and tr[770, 512] == 1 and tr[770, 1:512].sum() == 0
    assert tr[771, 511] == 1 and tr[771, 1023] == 1 and tr[771, :511].sum() == 0
    assert (tr[775, :512] == 0).all() and (tr[775, 512:1024] == 1).all()
    assert (tr[780, :65536] == np.arange(65536)).all()
    assert int(tr[779].sum()) == 450 * rows        # every range-checked cell counted once
    assert (tr[64:514] < 65536).all()
    alphas = np.array([0x1234567887654321 % P, 0x0fedcba998765432 % P], dtype=np.uint64)
    accs = np.zeros(2, np.uint64)
    rowsT = np.ascontiguousarray(tr.T)
    for r in list(range(0, 8)) + [510, 511, 512, 513, 1022, 1023, 1024, 40000]:
        loc, nxt = rowsT[r], rowsT[(r + 1) % rows]
        # z_last / lagrange selectors for a non-last, non-first row of the subgroup: the transition factor is
        # non-zero, first/last-row selectors vanish
        n = oracle.orc_g1_eval_constraints(oracle_lib.ptr(loc), oracle_lib.ptr(nxt), oracle_lib.ptr(alphas), 7, 0, 0,
                                           oracle_lib.ptr(accs))
        assert n == 1111
        assert accs[0] == 0 and accs[1] == 0, f"row {r}"
    # a corrupted cell must break it
    loc = rowsT[3].copy()
    loc[130] ^= np.uint64(1)
    oracle.orc_g1_eval_constraints(oracle_lib.ptr(loc), oracle_lib.ptr(rowsT[4]), oracle_lib.ptr(alphas), 7, 0, 0, oracle_lib.ptr(accs))
    assert accs[0] != 0
    for i in range(2):
        exp = synth.g1_scalar_mul_offset(synth.words_to_int(s[i]), (synth.words_to_int(x[i, :4]), synth.words_to_int(x[i, 4:])),
                                         (synth.words_to_int(o[i, :4]), synth.words_to_int(o[i, 4:])))
        assert (synth.words_to_int(outs[i, :4]), synth.words_to_int(outs[i, 4:])) == exp


@pytest.mark.parametrize("kind,name,n_constraints", [(1, "g2", 1693), (2, "fq", 770)])
def test_g2_and_fq_traces_satisfy_their_airs(oracle, kind, name, n_constraints):
    """G2 scalar-mul (W 1295, 1693 constraints) and Fq-exp (W 427, 770 constraints): generated traces satisfy every
    constraint on consecutive rows, outputs equal independent big-int arithmetic (the reference asserts the same against
    ark-bn254: g2/scalar_mul_stark.rs:105-108, exp_stark.rs:98-100)."""
    g = synth.G2_GEN
    assert synth.f2_mul(g[1], g[1]) == synth.f2_add(synth.f2_mul(g[0], synth.f2_mul(g[0], g[0])), synth.G2_B)
    if kind == 1:
        s, x, o = synth.g2_inputs(2)
        o[1] = x[1]
    else:
        (s, x), o = synth.fq_inputs(2), None
    tr, outs = oracle_lib.generate_trace(oracle, kind, s, x, o)
    assert tr.shape[0] == {1: 1295, 2: 427}[kind]
    rowsT = np.ascontiguousarray(tr.T)
    alphas = np.array([123456789123456789 % P, 987654321987654321 % P], dtype=np.uint64)
    accs = np.zeros(2, np.uint64)
    for r in [0, 1, 2, 3, 510, 511, 512, 513, 1023, 1024, 30000]:
        n = oracle.orc_eval_constraints(kind, oracle_lib.ptr(rowsT[r]), oracle_lib.ptr(rowsT[(r + 1) % tr.shape[1]]),
                                        oracle_lib.ptr(alphas), 7, 0, 0, oracle_lib.ptr(accs))
        assert n == n_constraints
        assert accs[0] == 0 and accs[1] == 0, (name, r)
    for i in range(2):
        si = synth.words_to_int(s[i])
        if kind == 1:
            assert synth.g2_from_words(outs[i]) == synth.g2_scalar_mul_offset(si, synth.g2_from_words(x[i]), synth.g2_from_words(o[i]))
        else:
            assert synth.words_to_int(outs[i]) == pow(synth.words_to_int(x[i]), si, synth.P)


def test_oracle_reproduces_the_committed_proof_digest(oracle):
    """tests/golden/proof_digests.json pins the oracle's whole Fq-exp proof transcript (the GPU proofs are compared with the
    same file in tests/test_gpu_golden.py): a change in the oracle cannot silently move the goal posts."""
    import hashlib
    import json
    import os
    from tools import synth
    from tests import oracle_lib
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proof_digests.json")))["fq_exp"]
    s, x = synth.fq_inputs(g["n"], seed=g["seed"])
    proof, outs, _, degree_bits = oracle_lib.prove(oracle, 2, s, x)
    assert degree_bits == g["degree_bits"] and proof.size == g["n_words"]
    assert hashlib.sha256(np.ascontiguousarray(proof, dtype="<u8").tobytes()).hexdigest() == g["sha256_proof_words"]


def test_committed_fork_fixture_is_what_the_oracle_proves_and_diff_tool_localises(oracle, tmp_path):
    """tests/golden/fixture/fq_expected.txt (tools/gen_fixture_expected.py) re-derived by the oracle; tools/diff_fixture.py (the
    pure-Python half of the fork cross-check, rust/README.md) accepts a dump in the Rust dumper's format, tolerates another
    valid proof-of-work witness and localises a witness difference to the trace and a transcript difference to its proof word."""
    import os
    from tools import compare_fixture as cf
    from tools import diff_fixture as df
    from tools import export_fixture_inputs as ex
    from tools import gen_fixture_expected as gen
    d = os.path.join(os.path.dirname(__file__), "golden", "fixture")
    s, x, o, text, trace, words = gen.make("fq", oracle)
    assert text == open(os.path.join(d, "fq_expected.txt")).read()
    s2, x2, _ = ex.read(os.path.join(d, "fq_inputs.txt"), "fq")
    assert np.array_equal(s, s2) and np.array_equal(x, x2)
    for kind in ("g1", "g2"):     # the other two kinds: inputs file = the synthetic inputs, expectation well-formed
        si, xi, oi = ex.inputs(kind, gen.CASES[kind], gen.SEED)
        sr, xr, orr = ex.read(os.path.join(d, kind + "_inputs.txt"), kind)
        assert np.array_equal(si, sr) and np.array_equal(xi, xr) and np.array_equal(oi, orr)
        e = df.parse_expected(os.path.join(d, kind + "_expected.txt"))
        w, a = gen.WIDTH_AUX[kind]
        assert e["ncols"] == w and e["nrows"] == 65536 and len(e["sections"]["head"][1]) == 192 + 4 * (w + a) + 12 + 192
        assert e["sections"]["init_challenger_state"][0] == e["n_words"] - 12
    dump = str(tmp_path / "dump.txt")
    exp = os.path.join(d, "fq_expected.txt")
    cf.write_dump(dump, trace, words)
    assert df.diff(exp, dump, out=lambda *_: None)
    other = words.copy()
    other[-13] += np.uint64(12345)                  # another witness: legitimate (rayon find_any), still "identical where it must be"
    cf.write_dump(dump, trace, other)
    log = []
    assert df.diff(exp, dump, out=log.append) and any("find_any" in ln for ln in log)
    bad_t = trace.copy()
    bad_t[5, 9] ^= np.uint64(1)
    cf.write_dump(dump, bad_t, words)
    log = []
    assert not df.diff(exp, dump, out=log.append) and any("trace: DIFFERENT" in ln and "[5]" in ln for ln in log)
    bad_w = words.copy()
    bad_w[200] ^= np.uint64(1)                      # an opening
    cf.write_dump(dump, trace, bad_w)
    log = []
    assert not df.diff(exp, dump, out=log.append) and any("head: DIFFERENT at proof word 200" in ln for ln in log)
