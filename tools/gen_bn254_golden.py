#!/usr/bin/env python3
"""Golden vectors for the BN254 layer, generated with Python big integers (independent of the oracle and
of the HIP build).  Pins Fq mul/inv and affine G1 add/double used by the trace generators against textbook
arithmetic, the check the reference performs against ark-bn254 at scalar_mul_stark.rs:105-108."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth

P = synth.P
rng = synth.Xoshiro256ss(2024)
vec = {"p": str(P), "fq_mul": [], "fq_inv": [], "g1_add": [], "g1_scalar_mul_offset": []}
for _ in range(8):
    a, b = rng.next_u256() % P, rng.next_u256() % P
    vec["fq_mul"].append([str(a), str(b), str(a * b % P)])
    vec["fq_inv"].append([str(a), str(pow(a, -1, P))])
vec["fq_mul"].append([str(P - 1), str(P - 1), "1"])
for i in range(6):
    k1, k2 = rng.next_u256() % synth.R_ORDER + 1, rng.next_u256() % synth.R_ORDER + 1
    a, b = synth.g1_mul(k1, synth.G1_GEN), synth.g1_mul(k2, synth.G1_GEN)
    if i == 0:
        b = a  # doubling branch
    c = synth.g1_add(a, b)
    vec["g1_add"].append([[str(a[0]), str(a[1])], [str(b[0]), str(b[1])], [str(c[0]), str(c[1])]])
for i in range(3):
    s = rng.next_u256()
    k1, k2 = rng.next_u256() % synth.R_ORDER + 1, rng.next_u256() % synth.R_ORDER + 1
    x, off = synth.g1_mul(k1, synth.G1_GEN), synth.g1_mul(k2, synth.G1_GEN)
    out = synth.g1_scalar_mul_offset(s, x, off)
    vec["g1_scalar_mul_offset"].append([str(s), [str(x[0]), str(x[1])], [str(off[0]), str(off[1])], [str(out[0]), str(out[1])]])
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bn254_kat.json")
json.dump(vec, open(out, "w"), indent=1)
print("wrote", out)
