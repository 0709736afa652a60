"""Soak: many batches in one process; free device memory and host RSS must stay flat, every proof must stay identical."""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth
import torch

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = pk.Context(0)
s, x, o = synth.g1_inputs(128 * 8)
ref = [p.words.copy() for p in ctx.prove_g1_batch(s, x, o)]


def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


free0, rss0, t0 = torch.cuda.mem_get_info(0)[0], rss_mb(), time.time()
for b in range(n_batches):
    proofs = ctx.prove_g1_batch(s, x, o)
    if b % 10 == 0:
        for p, r in zip(proofs, ref):
            assert np.array_equal(p.words, r)
        print(f"batch {b}: free device memory {torch.cuda.mem_get_info(0)[0] / 2**30:.2f} GiB, host RSS {rss_mb():.0f} MiB", flush=True)
    del proofs
dt = time.time() - t0
free1, rss1 = torch.cuda.mem_get_info(0)[0], rss_mb()
print(f"{n_batches} batches, {8 * n_batches / dt:.1f} proofs/s; device memory delta {(free0 - free1) / 2**20:.1f} MiB, host RSS delta {rss1 - rss0:.0f} MiB")
assert abs(free0 - free1) < 64 * 2**20
