"""One tall proof covering 1024 G1 scalar multiplications (N = 2^19), timed; compare with 8 x 128-instance proofs."""
import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from tools import synth
ctx = pk.Context(0)
s, x, o = synth.g1_inputs(1024)
ctx.prove_g1(s, x, o)
t0 = time.time()
n = 3
for _ in range(n):
    pr = ctx.prove_g1(s, x, o)
dt = (time.time() - t0) / n
print(f"tall proof 2^{pr.degree_bits}: {dt*1e3:.1f} ms per proof = {1024/dt:.0f} scalar-muls/s; stages", {k: round(v, 1) for k, v in pr.stage_ms.items()})
