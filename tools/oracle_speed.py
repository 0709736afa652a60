"""Single-core speed of the CPU oracle's two Poseidon permutations and wall time of a commitment (test infrastructure).
usage: python tools/oracle_speed.py"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests import oracle_lib

lib = oracle_lib.load()
lib.orc_permute_chain.argtypes = [oracle_lib.VP, oracle_lib.C.c_size_t, oracle_lib.C.c_int]
st = np.arange(12, dtype=np.uint64)
for fast in (0, 1):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        lib.orc_permute_chain(oracle_lib.ptr(st), 200000, fast)
        best = min(best, time.perf_counter() - t)
    print("permutation (%s): %.2f us" % ("sparse partial rounds" if fast else "textbook", best / 200000 * 1e6))
vals = np.random.default_rng(0).integers(0, 2**63, size=(64, 65536), dtype=np.uint64)
best = min((lambda t: (oracle_lib.commit_values(lib, vals), time.perf_counter() - t)[1])(time.perf_counter()) for _ in range(3))
print("from_values of 64 columns x 2^16 rows: %.2f s on %d threads" % (best, lib.orc_num_threads()))
