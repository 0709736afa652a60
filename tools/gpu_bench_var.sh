# run-to-run spread of the default bench line on one box: short and longer runs alternating
cd $GRAFT_REPO_ROOT
for cfg in "10 2" "20 5" "10 2" "20 5" "10 2"; do
  set -- $cfg
  echo -n "steps $1 warmup $2: "
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'seq', d['sequential_steps']['value'], 'ntt', d['roofline']['ms'], d['roofline']['valu_floor'].get('ntt_stage_clock_mhz'))"
done
