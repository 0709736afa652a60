# bench.py headline for deeper pipelines (FIFO admission); tuning only.
cd $GRAFT_REPO_ROOT
run() {
  d=$1; shift
  echo -n "depth $d $* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline --steps-in-flight $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ntt_ms', d['roofline']['ms'], 'excl', d['roofline']['exclusive']['ms'], d['roofline'].get('valu_floor'))"
}
run 3 BN254S_SLOTS=24
run 4 BN254S_SLOTS=32
run 4 BN254S_SLOTS=32 GPU_MAX_HW_QUEUES=32
run 3 BN254S_SLOTS=24 GPU_MAX_HW_QUEUES=24
run 3 BN254S_SLOTS=24 GPU_MAX_HW_QUEUES=8
run 6 BN254S_SLOTS=48
run 3 BN254S_SLOTS=24 BN254S_BIG_COST_EXCL=2
run 4 BN254S_SLOTS=24
run 3 BN254S_SLOTS=24
