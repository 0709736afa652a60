# Sweep of the scheduling knobs (tuning only): big-kernel semaphore, slots, hardware queues.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
L=gpurun_out/sweep/run.log
: > $L
run() { echo "== $*" | tee -a $L; env "$@" python tools/run_proofs.py 10 batch 2>&1 | tail -1 | tee -a $L; }
run BN254S_SLOTS=6
run BN254S_SLOTS=6 BN254S_BIG_COST_NTT=2
run BN254S_SLOTS=6
run BN254S_SLOTS=6 BN254S_BIG_COST_NTT=2
run BN254S_SLOTS=8 BN254S_BIG_COST_NTT=2
run BN254S_SLOTS=6 BN254S_BIG_CAP=4 BN254S_BIG_COST_NTT=3 BN254S_BIG_COST_EXCL=2
run BN254S_SLOTS=8 BN254S_BIG_CAP=4 BN254S_BIG_COST_NTT=3 BN254S_BIG_COST_EXCL=2
