"""One proof of a kind, a few times (tuning only): run under `rocprofv3 --kernel-trace` and feed the trace to
tools/proof_timeline.py to see the kernels of the LAST proof in order.  usage: python3 tools/one_proof.py g1|g2|fq [repeats]"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from tools import synth

kind = {"g1": 0, "g2": 1, "fq": 2}[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ins = (synth.g1_inputs, synth.g2_inputs, synth.fq_inputs)[kind](128)
ctx = pk.Context(0)
for _ in range(reps):
    p = ctx.prove_batch(kind, ins[0], ins[1], ins[2] if len(ins) > 2 else None, per_proof=128)[0]
print({k: round(v, 2) for k, v in p.stage_ms.items()})
