"""Compares a reference proof dumped by rust/dump_fixture.rs with this library's proof of the same inputs (GPU needed).
usage: python tools/compare_fixture.py fixture_inputs.txt fixture_proof.txt"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk

lines = open(sys.argv[1]).read().split("\n")
n = int(lines[0])
rows = np.array([[int(h, 16) for h in lines[1 + k].split()] for k in range(n)], dtype=np.uint64)
s, x, o = (np.ascontiguousarray(rows[:, a:b]) for a, b in ((0, 4), (4, 12), (12, 20)))
ref = open(sys.argv[2]).read().split()
ref = np.array([int(h, 16) for h in ref[1:1 + int(ref[0])]], dtype=np.uint64)
ctx = pk.Context(0)
pr = ctx.prove_g1(s, x, o)
got = pr.words
assert got.size == ref.size, (got.size, ref.size)
W, A = 781, 456
head = 192 + 4 * W + 4 * A + 4 + 8 + 64 * 3            # caps, openings, FRI caps
tail = 12                                              # init_challenger_state
final_poly = slice(got.size - 13 - 32, got.size - 13)
ok_head = np.array_equal(got[:head], ref[:head])
ok_final = np.array_equal(got[final_poly], ref[final_poly])
ok_state = np.array_equal(got[-tail:], ref[-tail:])
same_pow = got[-13] == ref[-13]
print("caps + openings + FRI caps:", "identical" if ok_head else "DIFFERENT at word %d" % int(np.flatnonzero(got[:head] != ref[:head])[0]))
print("final polynomial:", "identical" if ok_final else "DIFFERENT")
print("init_challenger_state:", "identical" if ok_state else "DIFFERENT")
print("pow witness: ours %d, reference %d (%s)" % (int(got[-13]), int(ref[-13]),
      "same: the query rounds must match too" if same_pow else "upstream's search is not deterministic; queries follow from it"))
if same_pow:
    print("query rounds:", "identical" if np.array_equal(got, ref) else "DIFFERENT")
# the reference proof itself must pass this library's verifier
ctx.verify(0, ref, pr.degree_bits, s, x, o, pr.outputs)
print("reference proof accepted by bn254s_verify")
sys.exit(0 if (ok_head and ok_final and ok_state) else 1)
