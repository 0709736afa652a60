set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "kernels or prove or g2_fq or verify or golden or robustness" > gpurun_out/pytest_p13.log 2>&1 || { tail -40 gpurun_out/pytest_p13.log; exit 1; }
tail -3 gpurun_out/pytest_p13.log
python tools/bench_hash.py sweep 2>&1 | tail -6
bash tools/gpu_lib_ab.sh libbn254stark_pre_sflag.so
