# Slots sweep with two / three steps open (tuning only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipesweep
L=gpurun_out/pipesweep/run2.log
: > $L
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['sequential_steps']['value'], d['roofline']['ms'])"; }
for cfg in "12 16 2" "14 16 2" "16 16 2" "16 24 2" "20 24 3" "12 16 3" "16 16 3" "24 32 3"; do
  set -- $cfg
  echo "== slots $1 hwq $2 steps-in-flight $3" | tee -a $L
  EXTRA="--steps-in-flight $3" BN254S_SLOTS=$1 GPU_MAX_HW_QUEUES=$2 run | tee -a $L
done
