# Where the wave cycles of the NTT kernels go (tuning only): SQ counters of tools/pmc_ntt.py
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/nttsq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d gpurun_out/nttsq -o a --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/nttsq/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d gpurun_out/nttsq -o b --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/nttsq/b.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/nttsq -o c --output-format csv -- python3 tools/pmc_ntt.py > gpurun_out/nttsq/c.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/nttsq/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "ntt" not in k and "copy" not in k: continue
    print(k)
    w = sum(v.get("SQ_WAVES", [1])) / max(len(v.get("SQ_WAVES", [1])), 1)
    for c, vals in sorted(v.items()):
        a = sum(vals) / len(vals)
        print("   %-24s %14.0f   per wave %10.1f" % (c, a, a / w))
PY
