"""Very tall single proofs (N = 2^21, 2^22: R = 32 / 64 blocks per column; 2^23: the radix-2 level above them, 16384 instances):
Fq-exp (light columns) and, optionally, G1.
usage: python tools/run_very_tall.py [fq|g1|g2] [log_rows=21]   Every proof is checked with bn254s_verify."""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth

kind = sys.argv[1] if len(sys.argv) > 1 else "fq"
log_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 21
n = (1 << log_rows) // 512
ctx = pk.Context(0)
if kind == "fq":
    s, x = synth.fq_inputs(n)
    o = None
    k = 2
elif kind == "g2":
    base = synth.g2_inputs(64)   # python G2 arithmetic is slower still: 64 distinct instances, tiled
    s, x, o = (np.tile(a, (n // 64, 1)) for a in base)
    k = 1
else:
    base = synth.g1_inputs(256)  # python EC arithmetic is slow: 256 distinct instances, tiled
    s, x, o = (np.tile(a, (n // 256, 1)) for a in base)
    k = 0
t0 = time.time()
prove = {"fq": lambda: ctx.prove_fq_exp(s, x), "g1": lambda: ctx.prove_g1(s, x, o), "g2": lambda: ctx.prove_g2(s, x, o)}[kind]
pr = prove()
t1 = time.time()
pr2 = prove()
t2 = time.time()
assert np.array_equal(pr.words, pr2.words)
print(f"{kind} 2^{pr.degree_bits} rows, {n} instances: {1e3 * (t2 - t1):.0f} ms (first call {1e3 * (t1 - t0):.0f} ms), "
      f"{pr.words.size} proof words; stages", {a: round(b, 1) for a, b in pr.stage_ms.items()}, flush=True)
ctx.verify(k, pr.words, pr.degree_bits, s, x, o, pr.outputs)
pk.verify_host(k, pr.words, pr.degree_bits, s, x, o, pr.outputs)
bad = pr.words.copy()
bad[64 * 3 + 7] ^= np.uint64(1)
try:
    ctx.verify(k, bad, pr.degree_bits, s, x, o, pr.outputs)
    raise SystemExit("corrupted proof accepted")
except pk.VerifyError as e:
    print("verified; corrupted proof rejected:", e, flush=True)
