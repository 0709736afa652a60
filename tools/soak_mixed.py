"""Mixed soak (tuning / validation only): one context alternates kinds, batch sizes and trace heights (plain and compact
workspaces); every proof is verified with the library's GPU verifier, the tall ones also with the host verifier."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth
import torch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = pk.Context(0)
g1 = synth.g1_inputs(256)
g2 = synth.g2_inputs(32)
fq = synth.fq_inputs(512)
free0 = torch.cuda.mem_get_info(0)[0]
t0 = time.time()
n = 0
for r in range(rounds):
    for kind, ins, per in ((0, g1, 128), (2, fq, 128), (1, g2, 32)):
        off = ins[2] if len(ins) > 2 else None
        proofs = ctx.prove_batch(kind, ins[0], ins[1], off, per_proof=per)
        for i, p in enumerate(proofs):
            lo, hi = per * i, min(per * (i + 1), ins[0].shape[0])
            ctx.verify(kind, p.words, p.degree_bits, ins[0][lo:hi], ins[1][lo:hi], None if off is None else off[lo:hi], p.outputs)
            n += 1
    # tall: Fq-exp at 2^19 (plain) and G1 at 2^17 through the compact workspace
    s, x = (np.tile(a, (2, 1)) for a in fq)
    p = ctx.prove_fq_exp(s, x)
    assert p.degree_bits == 19
    ctx.verify(2, p.words, 19, s, x, None, p.outputs)
    pk.verify_host(2, p.words, 19, s, x, None, p.outputs)
    import os
    os.environ["BN254S_FORCE_LOWMEM"] = "1"
    p = ctx.prove_g1(g1[0][:200], g1[1][:200], g1[2][:200])
    del os.environ["BN254S_FORCE_LOWMEM"]
    assert p.degree_bits == 17
    ctx.verify(0, p.words, 17, g1[0][:200], g1[1][:200], g1[2][:200], p.outputs)
    n += 2
    print(f"round {r}: {n} proofs verified, free device memory {torch.cuda.mem_get_info(0)[0] / 2**30:.1f} GiB, {time.time() - t0:.0f} s", flush=True)
print("ok")
