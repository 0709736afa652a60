set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/one
rocprofv3 --kernel-trace --stats -d gpurun_out/one -o one --output-format csv -- python3 tools/run_proofs.py 3 single > gpurun_out/one/run.log 2>&1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/one/one_kernel_trace.csv")))
import collections
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if "merkle" in n or "leaf_hash" in n:
        acc[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(k, len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
