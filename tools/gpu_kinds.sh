set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/kinds
python tools/run_kinds.py g1,g2,fq 3 | tee gpurun_out/kinds/run.log
rocprofv3 --kernel-trace --stats -d gpurun_out/kinds -o k --output-format csv -- python3 tools/run_kinds.py g2,fq 2 > gpurun_out/kinds/prof.log 2>&1
