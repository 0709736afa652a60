import sys, time
sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth
ctx = pk.Context(0)
s, x, o = synth.g1_inputs(128 * 8)
for k in (8, 16, 32):
    S, X, O = (np.tile(a, (k // 8, 1)) for a in (s, x, o))
    ctx.prove_g1_batch(S, X, O)
    t0 = time.time(); n = 3
    for _ in range(n): ctx.prove_g1_batch(S, X, O)
    dt = (time.time() - t0) / n
    print(f"batch of {k} proofs: {dt*1e3:.1f} ms, {k/dt:.2f} proofs/s", flush=True)
