"""Where a bench step's wall time goes (tuning only): the C call alone, the Python wrappers around it, HW-queue settings.
usage: python tools/step_overheads.py [steps=10]"""
import ctypes as C
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import numpy as np
import plonky2_bn254_amd as pk
from tools import synth
from plonky2_bn254_amd.lib import _ptr, default_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = pk.Context(0)
s, x, o = synth.g1_inputs(1024)
for _ in range(2):
    ctx.prove_g1_batch(s, x, o)
params = default_params()
lib = ctx._lib
t_c = t_wrap = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    outs = (C.c_void_p * 8)()
    a = time.perf_counter()
    rc = lib.bn254s_prove_g1_batch(ctx._h, C.byref(params), _ptr(s), _ptr(x), _ptr(o), 1024, 128, outs)
    b = time.perf_counter()
    assert rc == 0
    proofs = [pk.lib.Proof(lib, C.c_void_p(outs[i])) for i in range(8)]
    caps = np.stack([p.words[:192] for p in proofs])
    c = time.perf_counter()
    t_c += b - a
    t_wrap += c - b
dt = time.perf_counter() - t0
print(f"{steps} steps: {dt / steps * 1e3:.2f} ms per step = {8 * steps / dt:.2f} proofs/s; C call {t_c / steps * 1e3:.2f} ms, "
      f"Python wrappers {t_wrap / steps * 1e3:.2f} ms per step")
