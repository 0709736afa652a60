"""Writes synthetic inputs (plonky2_bn254_amd.synth, fixed seed) in the text format rust/shim/dump_fixture.rs reads.
usage: python tools/export_fixture_inputs.py [--kind g1|g2|fq] [n=128] [out=fixture_inputs.txt] [seed]"""
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools import synth

KINDS = {"g1": 0, "g2": 1, "fq": 2}


def inputs(kind: str, n: int, seed: int):
    """(scalars, x, offset or None) of `kind` in the ABI wire format."""
    if kind == "g1":
        return synth.g1_inputs(n, seed=seed)
    if kind == "g2":
        return synth.g2_inputs(n, seed=seed)
    s, x = synth.fq_inputs(n, seed=seed)
    return s, x, None


def write(path, s, x, o):
    with open(path, "w") as f:
        f.write("%d\n" % s.shape[0])
        for k in range(s.shape[0]):
            row = list(s[k]) + list(x[k]) + (list(o[k]) if o is not None else [])
            f.write(" ".join("%016x" % int(w) for w in row) + "\n")


def read(path, kind: str):
    import numpy as np
    lines = open(path).read().split("\n")
    n = int(lines[0])
    rows = np.array([[int(h, 16) for h in lines[1 + k].split()] for k in range(n)], dtype=np.uint64)
    pw = {"g1": 8, "g2": 16, "fq": 4}[kind]
    s, x = np.ascontiguousarray(rows[:, :4]), np.ascontiguousarray(rows[:, 4:4 + pw])
    o = None if kind == "fq" else np.ascontiguousarray(rows[:, 4 + pw:4 + 2 * pw])
    return s, x, o


if __name__ == "__main__":
    args = sys.argv[1:]
    kind = "g1"
    if args and args[0] == "--kind":
        kind, args = args[1], args[2:]
    n = int(args[0]) if len(args) > 0 else 128
    out = args[1] if len(args) > 1 else "fixture_inputs.txt"
    seed = int(args[2], 0) if len(args) > 2 else 0xF1C5
    write(out, *inputs(kind, n, seed))
    print("wrote", out, "(kind %s, seed 0x%X)" % (kind, seed))
