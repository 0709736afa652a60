# Scheduler knob sweep (tuning only): proofs/s of batches of 8 under different semaphore costs / slot counts.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/locksweep
L=gpurun_out/locksweep/run.log
: > $L
python tools/run_proofs.py 6 batch 2>&1 | tail -1 | tee -a $L
for cfg in "6 6 3 3 8" "6 4 3 3 8" "6 6 3 2 8" "6 3 3 3 8" "6 6 3 3 10" "6 6 2 3 8" "6 6 4 3 8" "6 6 3 3 6" "6 6 2 2 8" "12 12 6 5 8"; do
  set -- $cfg
  echo "== cap $1 ntt $2 excl $3 hash $4 slots $5" | tee -a $L
  BN254S_BIG_CAP=$1 BN254S_BIG_COST_NTT=$2 BN254S_BIG_COST_EXCL=$3 BN254S_BIG_COST_HASH=$4 BN254S_SLOTS=$5 python tools/run_proofs.py 6 batch 2>&1 | tail -1 | tee -a $L
done
