# bench.py headline with and without back-to-back admission of waiting NTT stages (BN254S_NTT_CONVOY); tuning only.
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
run BN254S_NTT_CONVOY=0
run BN254S_NTT_CONVOY=3
run BN254S_NTT_CONVOY=8
run BN254S_NTT_CONVOY=0
run BN254S_NTT_CONVOY=3
run BN254S_NTT_CONVOY=8
run BN254S_NTT_CONVOY=16
