"""NTT/LDE stage time as a function of the run length (tuning only): short runs are timed before the clocks settle."""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
ctx = pk.Context(0)
for it in (3, 10, 30, 100, 30, 3):
    print("iters %3d: %.3f ms per run" % (it, ctx.bench_ntt(1237, it)), flush=True)
