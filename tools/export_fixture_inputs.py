"""Writes the synthetic G1 inputs (plonky2_bn254_amd.synth, fixed seed) in the text format rust/dump_fixture.rs reads.
usage: python tools/export_fixture_inputs.py [n=128] [out=fixture_inputs.txt] [seed]"""
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from plonky2_bn254_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out = sys.argv[2] if len(sys.argv) > 2 else "fixture_inputs.txt"
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xF1C5
s, x, o = synth.g1_inputs(n, seed=seed)
with open(out, "w") as f:
    f.write("%d\n" % n)
    for k in range(n):
        f.write(" ".join("%016x" % int(w) for w in list(s[k]) + list(x[k]) + list(o[k])) + "\n")
print("wrote", out, "(seed 0x%X)" % seed)
