# Does an NTT stage that shares the GPU with one leaf hash keep its time when its waves have issue priority 3?  (tuning only)
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'])"
}
P=BN254S_LIB=$GRAFT_REPO_ROOT/tools/ubench/ab/libbn254stark_nttprio3.so
run A=1
run $P BN254S_BIG_COST_NTT=8
run $P BN254S_BIG_COST_NTT=10
run BN254S_BIG_COST_NTT=8
run $P BN254S_BIG_COST_NTT=8 BN254S_NTT_CONVOY=0
