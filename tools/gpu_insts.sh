# VALU instruction counts per kernel (SQ_INSTS_VALU) of single proofs: where the issue slots go.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/insts
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d gpurun_out/insts -o i --output-format csv -- python3 ${INSTS_CMD:-tools/run_proofs.py 2 single} > gpurun_out/insts/run.log 2>&1
python3 - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open("gpurun_out/insts/i_counter_collection.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[n] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for n, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:22]:
    print("%-34s launches %4d  VALU insts %12.0f  %5.1f%%  insts/wave %9.0f" % (n[:34], calls[n], v["SQ_INSTS_VALU"], 100 * v["SQ_INSTS_VALU"] / tot, v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))
PY
