"""Per-kind figures (G1 / G2 / Fq-exp): single-proof stage ms and batch-of-8 throughput.
usage: python tools/run_kinds.py [kinds=g1,g2,fq] [iters=3]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from tools import synth

kinds = (sys.argv[1] if len(sys.argv) > 1 else "g1,g2,fq").split(",")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = pk.Context(0)
# G2 base points cost ~10 ms each in python big-int arithmetic: 128 distinct instances, tiled 8x for the batch
gen = {"g1": (0, synth.g1_inputs), "g2": (1, synth.g2_inputs), "fq": (2, synth.fq_inputs)}
for name in kinds:
    kind, f = gen[name]
    ins = f(128)
    import numpy as np
    big = [np.tile(a, (8, 1)) for a in ins]
    off = ins[2] if len(ins) > 2 else None
    boff = big[2] if len(big) > 2 else None
    ctx.prove_batch(kind, ins[0], ins[1], off)
    acc = {}
    for i in range(iters):
        pr = ctx.prove_batch(kind, ins[0], ins[1], off)[0]
        for k, v in pr.stage_ms.items():
            acc[k] = acc.get(k, 0.0) + v / iters
    print(name, "single proof stage ms:", {k: round(v, 2) for k, v in acc.items()}, flush=True)
    ctx.prove_batch(kind, big[0], big[1], boff)
    t0 = time.time()
    for i in range(iters):
        ctx.prove_batch(kind, big[0], big[1], boff)
    dt = time.time() - t0
    print(f"{name} batch of 8 proofs (1024 instances): {dt / iters * 1e3:.1f} ms per batch, {8 * iters / dt:.2f} proofs/s, "
          f"{1024 * iters / dt:.0f} instances/s", flush=True)
