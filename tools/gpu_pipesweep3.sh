# Semaphore costs with twelve slots and two steps open (tuning only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipesweep
L=gpurun_out/pipesweep/run3.log
: > $L
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sequential_steps']['value'], d['roofline']['ms'])"; }
echo "== default" | tee -a $L; run | tee -a $L
for cfg in "9 9 3 2" "9 9 2 3" "12 12 3 3" "12 12 4 3" "12 12 3 4" "9 9 4 3" "9 9 3 4"; do
  set -- $cfg
  echo "== cap $1 ntt $2 excl $3 hash $4" | tee -a $L
  BN254S_BIG_CAP=$1 BN254S_BIG_COST_NTT=$2 BN254S_BIG_COST_EXCL=$3 BN254S_BIG_COST_HASH=$4 run | tee -a $L
done
