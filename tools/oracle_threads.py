"""Wall time of one 128-instance G1 proof by the CPU oracle for several OpenMP thread counts (run on the GPU box's host: the
figure bench.py reports as cpu_baseline; test infrastructure).  usage: python tools/oracle_threads.py [threads ...]"""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests import oracle_lib
from tools import synth

lib = oracle_lib.load()
s, x, o = synth.g1_inputs(128)
counts = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
print("host threads available:", os.cpu_count(), flush=True)
for n in counts:
    lib.orc_set_num_threads(n)
    t = time.time()
    _, _, tm, _ = oracle_lib.g1_prove(lib, s, x, o)
    print("%4d threads: %.2f s  stages %s" % (n, time.time() - t, " ".join("%.2f" % v for v in tm)), flush=True)
