set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python tools/run_proofs.py 5 single
python tools/run_proofs.py 3 batch
BN254S_SLOTS=1 python tools/run_proofs.py 3 batch
BN254S_SLOTS=8 python tools/run_proofs.py 3 batch
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 --output-format csv -- python3 tools/run_proofs.py 3 single > gpurun_out/prof/run.log 2>&1
ls -R gpurun_out/prof | head -20
