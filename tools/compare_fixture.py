"""Compares a reference dump written by rust/shim/dump_fixture.rs (trace column digests + proof words) with this library's
results for the same inputs (GPU needed).  A mismatch in the column digests localises a difference to trace generation, a
mismatch only in the proof words to a transcript convention.
usage: python tools/compare_fixture.py [--kind g1|g2|fq] fixture_inputs.txt fixture_proof.txt"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools.export_fixture_inputs import KINDS, read

DIGEST_K = 0x100000001B3


def column_digests(trace: np.ndarray) -> np.ndarray:
    """d_c = sum_i v[c][i] K^(N-1-i) mod 2^64 (what column_digest() in dump_fixture.rs folds)."""
    n = trace.shape[1]
    pw = np.empty(n, np.uint64)
    pw[0] = 1
    pw[1:] = DIGEST_K
    pw = np.cumprod(pw, dtype=np.uint64)[::-1]          # K^(N-1-i), wrapping
    return (trace * pw[None, :]).sum(axis=1, dtype=np.uint64)


def parse_dump(path):
    tok = open(path).read().split()
    ncols, nrows = int(tok[0]), int(tok[1])
    digests = np.array([int(h, 16) for h in tok[2:2 + ncols]], dtype=np.uint64)
    nw = int(tok[2 + ncols])
    words = np.array([int(h, 16) for h in tok[3 + ncols:3 + ncols + nw]], dtype=np.uint64)
    return nrows, digests, words


def write_dump(path, trace, words):
    """The same file format from Python (used to rehearse the comparison with an oracle-made dump)."""
    with open(path, "w") as f:
        f.write("%d %d\n" % trace.shape)
        for d in column_digests(trace):
            f.write("%016x\n" % int(d))
        f.write("%d\n" % words.size)
        for w in words:
            f.write("%016x\n" % int(w))


def compare(ctx, kind: str, inputs_path: str, dump_path: str, out=print) -> bool:
    k = KINDS[kind]
    s, x, o = read(inputs_path, kind)
    nrows, ref_digests, ref = parse_dump(dump_path)
    trace, _ = ctx.generate_trace(k, s, x, o)
    ok_trace = trace.shape == (ref_digests.size, nrows) and np.array_equal(column_digests(trace), ref_digests)
    if ok_trace:
        out("trace: all %d column digests identical" % ref_digests.size)
    else:
        bad = np.flatnonzero(column_digests(trace) != ref_digests) if trace.shape == (ref_digests.size, nrows) else []
        out("trace: DIFFERENT (shape %s vs %s, first columns %s)" % (trace.shape, (ref_digests.size, nrows), [int(c) for c in bad[:8]]))
    pr = {0: ctx.prove_g1, 1: ctx.prove_g2}[k](s, x, o) if k != 2 else ctx.prove_fq_exp(s, x)
    got = pr.words
    if got.size != ref.size:
        out("proof: DIFFERENT word count %d vs %d" % (got.size, ref.size))
        return False
    n_final = pr.section("final_poly").size
    head = got.size - 13 - n_final - pr.section("query_round_proofs").size       # caps, openings, FRI caps
    final_poly = slice(got.size - 13 - n_final, got.size - 13)
    ok_head = np.array_equal(got[:head], ref[:head])
    ok_final = np.array_equal(got[final_poly], ref[final_poly])
    ok_state = np.array_equal(got[-12:], ref[-12:])
    same_pow = got[-13] == ref[-13]
    out("caps + openings + FRI caps: " + ("identical" if ok_head else "DIFFERENT at word %d" % int(np.flatnonzero(got[:head] != ref[:head])[0])))
    out("final polynomial: " + ("identical" if ok_final else "DIFFERENT"))
    out("init_challenger_state: " + ("identical" if ok_state else "DIFFERENT"))
    out("pow witness: ours %d, reference %d (%s)" % (int(got[-13]), int(ref[-13]),
        "same: the query rounds must match too" if same_pow else "upstream's search is not deterministic; queries follow from it"))
    ok_queries = True
    if same_pow:
        ok_queries = bool(np.array_equal(got, ref))
        out("query rounds: " + ("identical" if ok_queries else "DIFFERENT"))
    ctx.verify(k, ref, pr.degree_bits, s, x, o, pr.outputs)     # the reference proof itself must pass this library's verifier
    out("reference proof accepted by bn254s_verify")
    return ok_trace and ok_head and ok_final and ok_state and ok_queries


if __name__ == "__main__":
    import plonky2_bn254_amd as pk
    args = sys.argv[1:]
    kind = "g1"
    if args and args[0] == "--kind":
        kind, args = args[1], args[2:]
    sys.exit(0 if compare(pk.Context(0), kind, args[0], args[1]) else 1)
