# Runs on the GPU box: full gpu test suite, smoke, bench, and the rocprofv3 kernel stats of the bench command.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
make -C oracle > /dev/null
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 3 --warmup 1 | tee gpurun_out/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof/bench_prof.log 2>&1
tail -2 gpurun_out/prof/bench_prof.log
