# Scheduler A/B on the GPU box: bench.py headline under several admission settings (tuning only).
# usage: bash tools/gpu_sched_ab.sh > gpurun_out/sched_ab.txt
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 12 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
run BN254S_SCHED_FIFO=1
run BN254S_SCHED_FIFO=0
run BN254S_SCHED_FIFO=1
run BN254S_SCHED_FIFO=0
run BN254S_SCHED_FIFO=1 BN254S_BIG_COST_HASH=4
run BN254S_SCHED_FIFO=1 BN254S_BIG_COST_HASH=4 BN254S_BIG_COST_EXCL=2
run BN254S_SCHED_FIFO=1 BN254S_BIG_COST_EXCL=2
run BN254S_SCHED_FIFO=1 BN254S_BIG_COST_NTT=6
run BN254S_SCHED_FIFO=0 BN254S_BIG_COST_NTT=6
run BN254S_SCHED_FIFO=1 BN254S_SLOTS=16
run BN254S_SCHED_FIFO=0 BN254S_SLOTS=16
run BN254S_SCHED_FIFO=1 BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12 BN254S_BIG_COST_HASH=4 BN254S_BIG_COST_EXCL=3
