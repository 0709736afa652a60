#!/usr/bin/env python3
"""Fork cross-check for a machine that has cargo but NO GPU and none of this repository's libraries (pure Python, no numpy).

Compares a dump written by the reference crate itself (rust/shim/dump_fixture.rs: the reference's own `generate_trace` +
`prove` on the committed inputs tests/golden/fixture/<kind>_inputs.txt) with the expectation committed next to it
(tests/golden/fixture/<kind>_expected.txt: what this repository's prover - GPU and CPU oracle, word for word the same -
produces on those inputs).  Everything up to the proof-of-work witness is deterministic and must be identical.

usage: python tools/diff_fixture.py tests/golden/fixture/g1_expected.txt g1_ref.txt      (same for g2, fq)
exit code 0 = identical where it must be.

Expected-file format (text, whitespace separated, words as 16 hex digits):
  kind <g1|g2|fq>  inputs <n>  seed <hex>
  trace <ncols> <nrows>   then ncols column digests (d_c = sum_i v[c][i] K^(nrows-1-i) mod 2^64, K = 0x100000001B3)
  words <total word count of the proof>
  then sections `section <name> <offset> <count>` each followed by count words: head (caps, openings, FRI caps),
  final_poly, pow_witness (smallest witness; upstream's rayon find_any may return another), init_challenger_state
"""
import sys


def parse_expected(path):
    tok = open(path).read().split()
    pos = 0

    def take(n=1):
        nonlocal pos
        out = tok[pos:pos + n]
        pos += n
        return out

    exp = {"sections": {}}
    while pos < len(tok):
        key = take()[0]
        if key in ("kind", "inputs", "seed"):
            exp[key] = take()[0]
        elif key == "trace":
            ncols, nrows = (int(v) for v in take(2))
            exp["ncols"], exp["nrows"] = ncols, nrows
            exp["digests"] = [int(h, 16) for h in take(ncols)]
        elif key == "words":
            exp["n_words"] = int(take()[0])
        elif key == "section":
            name, off, cnt = take(3)
            exp["sections"][name] = (int(off), [int(h, 16) for h in take(int(cnt))])
        else:
            raise SystemExit("unexpected token %r in %s" % (key, path))
    return exp


def parse_dump(path):
    """rust/shim/dump_fixture.rs: `ncols nrows`, ncols digests, `n_words`, the proof words."""
    tok = open(path).read().split()
    ncols, nrows = int(tok[0]), int(tok[1])
    digests = [int(h, 16) for h in tok[2:2 + ncols]]
    nw = int(tok[2 + ncols])
    words = [int(h, 16) for h in tok[3 + ncols:3 + ncols + nw]]
    if len(words) != nw:
        raise SystemExit("%s: truncated dump (%d of %d words)" % (path, len(words), nw))
    return ncols, nrows, digests, words


def diff(expected_path, dump_path, out=print):
    exp = parse_expected(expected_path)
    ncols, nrows, digests, words = parse_dump(dump_path)
    ok = True
    if (ncols, nrows) != (exp["ncols"], exp["nrows"]):
        out("trace shape: DIFFERENT %dx%d, expected %dx%d" % (ncols, nrows, exp["ncols"], exp["nrows"]))
        ok = False
    else:
        bad = [c for c in range(ncols) if digests[c] != exp["digests"][c]]
        out("trace: all %d column digests identical" % ncols if not bad else
            "trace: DIFFERENT in %d columns, first %s  (a witness-generation difference, independent of every prover convention)"
            % (len(bad), bad[:8]))
        ok = ok and not bad
    if len(words) != exp["n_words"]:
        out("proof: DIFFERENT word count %d, expected %d" % (len(words), exp["n_words"]))
        return False
    for name in ("head", "final_poly", "init_challenger_state"):
        off, ref = exp["sections"][name]
        got = words[off:off + len(ref)]
        if got == ref:
            out("%s: identical (%d words)" % (name, len(ref)))
        else:
            first = next(i for i in range(len(ref)) if got[i] != ref[i])
            out("%s: DIFFERENT at proof word %d (a transcript convention: SURVEY.md App. A)" % (name, off + first))
            ok = False
    off, ref = exp["sections"]["pow_witness"]
    out("pow_witness: reference %d, expected smallest %d (%s)" % (words[off], ref[0],
        "same" if words[off] == ref[0] else "upstream searches with rayon find_any: a different valid witness is legitimate; the "
        "query rounds follow from it and are not compared"))
    return ok


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    sys.exit(0 if diff(sys.argv[1], sys.argv[2]) else 1)
