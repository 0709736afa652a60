# A/B on the GPU box (tuning only): host threads spinning vs sleeping while they wait for the GPU (BN254S_SYNC=blocking), and the
# issue priority of the NTT kernels (tools/ubench/ab/libbn254stark_nttprio3.so); bench.py headline, CPU seconds of the run.
cd $GRAFT_REPO_ROOT
TIMEFORMAT="   cpu %U+%S s, wall %R s"
run() {
  echo -n "$* : "
  time (env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt_ms', d['roofline']['ms'], 'excl', d['roofline']['exclusive']['ms'])")
}
run A=1
run BN254S_SYNC=blocking
run BN254S_LIB=$GRAFT_REPO_ROOT/tools/ubench/ab/libbn254stark_nttprio3.so
run A=2
run BN254S_SYNC=blocking
run BN254S_LIB=$GRAFT_REPO_ROOT/tools/ubench/ab/libbn254stark_nttprio3.so
