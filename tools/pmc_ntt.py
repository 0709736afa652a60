"""Driver for the PMC passes: the NTT/LDE stage on 1237 columns (one proof's trace + aux) and a calibration copy."""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk
ctx = pk.Context(0)
ctx.bench_copy(1237 * 65536, 3)      # 648 MB read + 648 MB written per launch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3   # 30: clocks settled (tools/bench_ntt_iters.py)
print("ntt ms", ctx.bench_ntt(1237, iters))
