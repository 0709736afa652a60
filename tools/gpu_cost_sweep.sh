# bench.py headline under different admission costs of the wide sections (FIFO admission, 32 slots, 4 steps in flight); tuning only.
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ntt_ms', d['roofline']['ms'])"
}
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12 BN254S_BIG_COST_EXCL=2
run BN254S_BIG_CAP=15 BN254S_BIG_COST_NTT=15
run BN254S_BIG_CAP=15 BN254S_BIG_COST_NTT=15 BN254S_BIG_COST_HASH=4
run BN254S_BIG_CAP=18 BN254S_BIG_COST_NTT=18
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12 BN254S_BIG_COST_HASH=4 BN254S_BIG_COST_EXCL=2
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12 BN254S_BIG_COST_HASH=5
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12
run BN254S_BIG_CAP=15 BN254S_BIG_COST_NTT=15
run A=1
