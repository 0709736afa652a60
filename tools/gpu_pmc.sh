set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc -o fetch --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc -o write --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/pmc/write.log 2>&1
ls gpurun_out/pmc
python3 - <<'PY'
import csv, glob, collections
for name in ("fetch","write"):
    f = glob.glob(f"gpurun_out/pmc/{name}_counter_collection.csv")
    if not f: print("missing", name); continue
    rows = list(csv.DictReader(open(f[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(name, k, "n=%d avg=%.1f" % (len(v), sum(v)/len(v)))
PY
