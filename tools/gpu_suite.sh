# Runs on the GPU box: full gpu test suite (progress in gpurun_out/pytest_gpu.log), smoke, a short bench run.
# usage: bash tools/gpu_suite.sh [pytest -k expression]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
if [ -n "$1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "$1" --durations=15 > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
else
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; exit 1; }
fi
tail -25 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
