# bench.py headline for steps-in-flight x slots (FIFO admission, the default); tuning only.
cd $GRAFT_REPO_ROOT
run() {
  d=$1; shift
  echo -n "depth $d $* : "
  env "$@" python bench.py --steps 12 --warmup 2 --no-extras --no-cpu-baseline --steps-in-flight $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ntt_ms', d['roofline']['ms'])"
}
for rep in 1 2; do
run 1 BN254S_SLOTS=8
run 1 BN254S_SLOTS=12
run 2 BN254S_SLOTS=12
run 2 BN254S_SLOTS=16
run 2 BN254S_SLOTS=20
run 3 BN254S_SLOTS=16
run 3 BN254S_SLOTS=24
done
