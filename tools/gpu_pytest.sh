# Runs on the GPU box: selected gpu tests with output shown (-s), log in gpurun_out/pytest_sel.log
# usage: bash tools/gpu_pytest.sh "<-k expression>"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s -k "$1" --durations=10 > gpurun_out/pytest_sel.log 2>&1
rc=$?
tail -60 gpurun_out/pytest_sel.log
exit $rc
