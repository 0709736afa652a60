import sys
sys.path.insert(0, "/root/repo")
import plonky2_bn254_amd as pk
from tools import synth
ctx = pk.Context(0)
for kind, name, ins in ((0, "g1", synth.g1_inputs(128)), (1, "g2", synth.g2_inputs(128)), (2, "fq", synth.fq_inputs(128))):
    off = ins[2] if len(ins) > 2 else None
    best = None
    for it in range(6):
        p = ctx.prove_batch(kind, ins[0], ins[1], off, per_proof=128)[0]
        if best is None or p.stage_ms["total"] < best["total"]:
            best = dict(p.stage_ms)
    print(name, {k: round(v, 2) for k, v in best.items()}, flush=True)
