"""BASELINE.json configs[4] on ONE GPU: map_to_g2 for n Fq2 inputs = 2n fq_exp jobs (Legendre symbols) + n G2
cofactor-clearing scalar multiplications, in 128-instance proofs, through bn254s_map_to_g2 (device front-end).
usage: python tools/run_config5.py [n=4096] [verify=0|1]
(the 8-GPU form shards the inputs across ranks exactly like bench.py: no data-path collective)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
from tools import map_to_g2_ref as m2g
from tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
check = len(sys.argv) > 2 and sys.argv[2] == "1"
ctx = pk.Context(0)
us = m2g.inputs(n)
u = np.array([synth._to_words(a[0]) + synth._to_words(a[1]) for a in us], dtype=np.uint64)
# random non-infinity offsets (set_random_g2): 128 distinct points, tiled (python G2 arithmetic is slow)
_, _, base_off = synth.g2_inputs(min(n, 128), seed=0x706C6F6E6B7932 + 5)
off = np.tile(base_off, ((n + base_off.shape[0] - 1) // base_off.shape[0], 1))[:n].copy()
ctx.map_to_g2(u[:2048], off[:2048])  # warm-up (tables, and the workspaces of all twelve slots for both kinds: 16 G2 proofs)
t0 = time.time()
out, fq_jobs, g2_jobs, pf, pg = ctx.map_to_g2(u, off)
dt = time.time() - t0
print(f"config 5, n = {n} Fq2 inputs -> {len(pf)} fq_exp proofs + {len(pg)} G2 proofs + device front-end: {dt * 1e3:.0f} ms end to end, "
      f"{n / dt:.0f} inputs/s, {(len(pf) + len(pg)) / dt:.1f} proofs/s", flush=True)
if check:
    fs, fx = np.ascontiguousarray(fq_jobs[:, :4]), np.ascontiguousarray(fq_jobs[:, 4:])
    gs, gx = np.ascontiguousarray(g2_jobs[:, :4]), np.ascontiguousarray(g2_jobs[:, 4:])
    for i, p in enumerate(pf):
        lo = 128 * i
        ctx.verify(2, p.words, p.degree_bits, fs[lo:lo + 128], fx[lo:lo + 128], None, p.outputs)
    for i, p in enumerate(pg):
        lo = 128 * i
        ctx.verify(1, p.words, p.degree_bits, gs[lo:lo + 128], gx[lo:lo + 128], off[lo:lo + 128], p.outputs)
    for k in range(min(n, 3)):
        legendre = [pow(synth.words_to_int(fx[2 * k + i]), (synth.P - 1) // 2, synth.P) for i in range(2)]
        pt = m2g.select_point(us[k], legendre[0] == 1, legendre[1] == 1)
        assert synth.g2_from_words(out[k]) == synth.g2_mul(m2g.COFACTOR, pt)
    print("all proofs verified (bn254s_verify); first images equal cofactor * point (python big-int check)", flush=True)
