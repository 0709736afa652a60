"""Small driver used under rocprofv3: warm-up proof, then `n` single proofs; prints per-stage GPU ms."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "single"
ctx = pk.Context(0)
s, x, o = synth.g1_inputs(128 * 8)
ctx.prove_g1(s[:128], x[:128], o[:128])
if mode == "single":
    acc = {}
    t0 = time.time()
    for i in range(n):
        pr = ctx.prove_g1(s[:128], x[:128], o[:128])
        for k, v in pr.stage_ms.items():
            acc[k] = acc.get(k, 0.0) + v / n
    dt = time.time() - t0
    print("avg stage ms:", {k: round(v, 3) for k, v in acc.items()})
    print(f"wall per proof {dt / n * 1e3:.2f} ms")
else:
    ctx.prove_g1_batch(s, x, o)
    t0 = time.time()
    for i in range(n):
        prs = ctx.prove_g1_batch(s, x, o)
    dt = time.time() - t0
    print(f"batch of 8 proofs: {dt / n * 1e3:.2f} ms per batch, {8 * n / dt:.2f} proofs/s")
