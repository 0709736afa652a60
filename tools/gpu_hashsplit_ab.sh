# bench.py headline with a proof's leaf hash admitted as 1 / 2 / 4 sections (BN254S_HASH_SPLIT); tuning only.
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
run BN254S_HASH_SPLIT=1
run BN254S_HASH_SPLIT=2
run BN254S_HASH_SPLIT=4
run BN254S_HASH_SPLIT=1
run BN254S_HASH_SPLIT=2
run BN254S_HASH_SPLIT=4
run BN254S_HASH_SPLIT=2 BN254S_BIG_COST_HASH=6
