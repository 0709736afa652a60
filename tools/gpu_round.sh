# Runs on the GPU box: selected tests, then bench twice with the default settings, then single-proof stage times.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/pytest_sel.log 2>&1 || { tail -40 gpurun_out/pytest_sel.log; exit 1; }
tail -3 gpurun_out/pytest_sel.log
python bench.py --steps 24 --warmup 4 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err || { tail -20 gpurun_out/bench_a.err; exit 1; }
cat gpurun_out/bench_a.json
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_b.err || { tail -20 gpurun_out/bench_b.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_default.json')); print('default run:', d['value'], d['ms_per_step'], d['config']['steps_in_flight'], d['sequential_steps'])"
python tools/stage_times.py > gpurun_out/stage_times.txt 2>&1; cat gpurun_out/stage_times.txt
