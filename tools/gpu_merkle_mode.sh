# bench.py headline for the Merkle upper-level kernel choices (tuning only): cooperative threshold, hand-scheduled level kernel.
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
run A=1
run BN254S_MERKLE_LEVEL_ASM=1
run BN254S_COOP_MAX_NODES=4096
run BN254S_COOP_MAX_NODES=1024
run BN254S_COOP_MAX_NODES=1024 BN254S_MERKLE_LEVEL_ASM=1
run BN254S_COOP_MAX_NODES=256 BN254S_MERKLE_LEVEL_ASM=1
run BN254S_COOP_MAX_NODES=0 BN254S_MERKLE_LEVEL_ASM=1
run A=2
run BN254S_MERKLE_LEVEL_ASM=1
run BN254S_COOP_MAX_NODES=1024 BN254S_MERKLE_LEVEL_ASM=1
