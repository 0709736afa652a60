cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "trace or g2_fq or robustness or prove" > gpurun_out/pytest_sel.log 2>&1 || { tail -40 gpurun_out/pytest_sel.log; exit 1; }
tail -3 gpurun_out/pytest_sel.log
bash tools/gpu_ntt_ab.sh > gpurun_out/ntt_ab.txt 2>&1; cat gpurun_out/ntt_ab.txt
python tools/stage_times.py > gpurun_out/stage_times.txt 2>&1; cat gpurun_out/stage_times.txt
python tools/oracle_threads.py 16 32 64 128 > gpurun_out/oracle_threads.txt 2>&1; cat gpurun_out/oracle_threads.txt
run() {
  echo -n "$* : "
  env "$@" python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'ntt_ms', d['roofline']['ms'])"
}
run A=1
run BN254S_BIG_COST_HASH=4
run BN254S_BIG_COST_EXCL=2
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12
run BN254S_BIG_CAP=12 BN254S_BIG_COST_NTT=12 BN254S_BIG_COST_HASH=4
run A=2
