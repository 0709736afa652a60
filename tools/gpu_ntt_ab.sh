# NTT/LDE stage time of the shipped library against two TIMING-ONLY builds (tools/ubench/ab/: -DBN254S_NTT_AB=1 drops the inner
# twiddles of the pass-1 tiles, =2 replaces them by power-of-two twiddles): what a 64 x 64 x 16 decomposition could save.
cd $GRAFT_REPO_ROOT
for lib in plonky2_bn254_amd/libbn254stark.so tools/ubench/ab/libbn254stark_ntt_ab1.so tools/ubench/ab/libbn254stark_ntt_ab2.so; do
python - $lib <<'PY'
import ctypes as C, sys
lib = C.CDLL(sys.argv[1])
h = C.c_void_p()
assert lib.bn254s_ctx_create(0, C.byref(h)) == 0
lib.bn254s_bench_ntt.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
ms = C.c_float()
best = 1e9
for rep in range(4):
    assert lib.bn254s_bench_ntt(h, 1237, 30, C.byref(ms)) == 0
    best = min(best, ms.value)
print("%-55s NTT/LDE stage, 1237 columns: %.4f ms (best of 4 x 30 iterations)" % (sys.argv[1], best))
PY
done
