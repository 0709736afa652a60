set -e
cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -15
python - <<'PY'
import plonky2_bn254_amd as pk
ctx = pk.Context(0)
for nc in (64, 256, 781, 1241):
    ms = ctx.bench_ntt(nc, 5)
    gb = 40*65536*nc/1e9
    print(f"ntt ncols={nc} {ms:.3f} ms  algorithmic {gb/ms*1e3:.0f} GB/s  frac {gb/ms*1e3/8000:.3f}")
PY
