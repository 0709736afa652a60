# bench.py headline under admission costs, with the NTT convoy and the Merkle throughput kernels in place (tuning only)
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
run A=1
run BN254S_BIG_COST_HASH=3
run BN254S_BIG_COST_EXCL=3
run BN254S_BIG_COST_HASH=3 BN254S_BIG_COST_EXCL=1
run BN254S_BIG_CAP=16 BN254S_BIG_COST_NTT=16
run BN254S_BIG_CAP=10 BN254S_BIG_COST_NTT=10
run BN254S_SLOTS=40 A=3
run A=2
run BN254S_NTT_CONVOY=8
