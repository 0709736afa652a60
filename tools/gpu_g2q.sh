# G2 / Fq quotient kernels after a change: their parity tests, then the bench line with the other kinds
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "g2 or fq or verify or stream or config" --durations=5 > gpurun_out/pytest_g2q.log 2>&1 || { tail -40 gpurun_out/pytest_g2q.log; exit 1; }
tail -8 gpurun_out/pytest_g2q.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_g2q.json 2> gpurun_out/bench_g2q.err || { tail -20 gpurun_out/bench_g2q.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/bench_g2q.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['other_kinds']))"
