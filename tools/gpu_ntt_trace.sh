set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/nt
rocprofv3 --kernel-trace --stats -d gpurun_out/nt -o nt --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/nt/log.txt 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/nt/nt_kernel_stats.csv")):
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
