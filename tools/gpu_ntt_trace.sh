set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ntt
rocprofv3 --kernel-trace --stats -d gpurun_out/ntt -o ntt --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/ntt/run.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/ntt/ntt_kernel_stats.csv")):
    print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
