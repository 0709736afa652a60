# NTT iteration loop (tuning only): field self test + kernel parity tests + stage time + kernel trace
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/nttchk
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5
python tools/bench_ntt_iters.py
rocprofv3 --kernel-trace --stats -d gpurun_out/nttchk -o ntt --output-format csv -- python3 tools/pmc_ntt.py 30 > gpurun_out/nttchk/run.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/nttchk/ntt_kernel_stats.csv")):
    if "ntt" in r["Name"] or "copy_u64" in r["Name"]: print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
