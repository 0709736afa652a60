"""BASELINE.json configs[4] on ONE GPU: map_to_g2 for n Fq2 inputs = 2n fq_exp jobs (Legendre symbols) + n G2
cofactor-clearing scalar multiplications, in 128-instance proofs.  usage: python tools/run_config5.py [n=4096] [verify=0|1]
(the 8-GPU form shards the proofs across ranks exactly like bench.py: no data-path collective)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from plonky2_bn254_amd import map_to_g2 as m2g
from plonky2_bn254_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
check = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = pk.Context(0)
t0 = time.time()
us = m2g.inputs(n)
fs, fx = m2g.fq_exp_jobs(us)
t_host1 = time.time() - t0
ctx.prove_batch(2, fs[:128], fx[:128])  # warm-up (tables, workspaces)
t0 = time.time()
pf = ctx.prove_batch(2, fs, fx)
t_fq = time.time() - t0
legendre = [synth.words_to_int(w) for p in pf for w in p.outputs.reshape(-1, 4)]
t0 = time.time()
gs, gx, goff, pts = m2g.g2_jobs(us, legendre)
t_host2 = time.time() - t0
ctx.prove_batch(1, gs[:128], gx[:128], goff[:128])
t0 = time.time()
pg = ctx.prove_batch(1, gs, gx, goff)
t_g2 = time.time() - t0
print(f"config 5, n = {n} Fq2 inputs: {len(pf)} fq_exp proofs in {t_fq * 1e3:.0f} ms ({len(pf) / t_fq:.1f} proofs/s), "
      f"{len(pg)} G2 proofs in {t_g2 * 1e3:.0f} ms ({len(pg) / t_g2:.1f} proofs/s); GPU pipeline {n / (t_fq + t_g2):.0f} inputs/s; "
      f"host big-int front-end {t_host1 + t_host2:.1f} s (python)", flush=True)
if check:
    for i, p in enumerate(pf):
        lo = 128 * i
        ctx.verify(2, p.words, p.degree_bits, fs[lo:lo + 128], fx[lo:lo + 128], None, p.outputs)
    for i, p in enumerate(pg):
        lo = 128 * i
        ctx.verify(1, p.words, p.degree_bits, gs[lo:lo + 128], gx[lo:lo + 128], goff[lo:lo + 128], p.outputs)
    outs = np.concatenate([p.outputs.reshape(-1, 16) for p in pg])
    for k in range(min(n, 4)):
        assert m2g.finish(outs[k], pts[k][1]) == synth.g2_mul(m2g.COFACTOR, pts[k][0])
    print("all proofs verified (bn254s_verify); first outputs equal cofactor * point", flush=True)
