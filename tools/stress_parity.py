"""Randomised differential test: GPU trace generation vs the CPU oracle over many seeds (all three STARKs), plus a few whole
proofs.  usage: python tools/stress_parity.py [n_seeds=40]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth
from tests import oracle_lib

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = oracle_lib.load()
ctx = pk.Context(0)
t0 = time.time()
bad = 0
for seed in range(1000, 1000 + n_seeds):
    rng = np.random.default_rng(seed)
    for kind, gen, n in ((0, synth.g1_inputs, 5), (2, synth.fq_inputs, 7), (1, synth.g2_inputs, 2 if seed % 4 == 0 else 0)):
        if n == 0:
            continue
        ins = list(gen(n, seed=seed))
        # sprinkle structured scalars: runs of ones / zeros, single bits, values around the group order
        for k in range(n):
            mode = int(rng.integers(0, 6))
            if mode == 0:
                ins[0][k] = synth._to_words((1 << int(rng.integers(1, 256))) - 1)
            elif mode == 1:
                ins[0][k] = synth._to_words(1 << int(rng.integers(0, 256)))
            elif mode == 2:
                ins[0][k] = synth._to_words((synth.R_ORDER + int(rng.integers(-3, 4))) % (1 << 256))
        off = ins[2] if len(ins) > 2 else None
        try:
            ref, ref_out = oracle_lib.generate_trace(lib, kind, ins[0], ins[1], off)
            ref_err = None
        except RuntimeError as e:
            ref_err = str(e)
        try:
            got, got_out = ctx.generate_trace(kind, ins[0], ins[1], off)
            got_err = None
        except RuntimeError as e:
            got_err = str(e)
        if (ref_err is None) != (got_err is None):
            bad += 1
            print("ERROR BEHAVIOUR DIFFERS", seed, kind, ref_err, got_err, flush=True)
        elif ref_err is None and not (np.array_equal(got, ref) and np.array_equal(got_out.reshape(ref_out.shape), ref_out)):
            bad += 1
            print("TRACE MISMATCH", seed, kind, np.argwhere(got != ref)[:3].tolist(), flush=True)
    if seed % 10 == 0:
        print(f"seed {seed} done, {time.time() - t0:.0f} s", flush=True)
# whole proofs on a few seeds (the oracle needs ~15 s each on the GPU box's cores)
for seed in (7001, 7002):
    s, x, o = synth.g1_inputs(3, seed=seed)
    ref, _, _, _ = oracle_lib.prove(lib, 0, s, x, o)
    got = ctx.prove_g1(s, x, o).words
    if not np.array_equal(ref, got):
        bad += 1
        print("PROOF MISMATCH", seed, flush=True)
    print(f"proof seed {seed} compared, {time.time() - t0:.0f} s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
