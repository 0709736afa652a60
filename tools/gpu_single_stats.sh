# Per-kernel average durations of single (un-overlapped) proofs: tuning aid.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/single
rocprofv3 --kernel-trace --stats -d gpurun_out/single -o s --output-format csv -- python3 tools/run_proofs.py 4 single > gpurun_out/single/run.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/single/s_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4:
        print("%-60s calls %4s avg us %9.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
