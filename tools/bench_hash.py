"""Poseidon leaf-hash throughput alone on the GPU: k_leaf_hash over 2^k leaves of 781 / 456 elements (98 / 57 permutations each)."""
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import plonky2_bn254_amd as pk

ctx = pk.Context(0)
cases = ((781, 17), (456, 17), (781, 20))
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    cases = tuple((781, k) for k in (15, 16, 17, 18, 19, 20))
for ncols, logl in cases:
    ms = ctx.bench_leafhash(ncols, logl, 5)
    perms = (1 << logl) * ((ncols + 7) // 8)
    print(f"leaf hash {ncols} cols x 2^{logl} leaves: {ms:.3f} ms, {perms / ms / 1e6:.3f} G perm/s", flush=True)
