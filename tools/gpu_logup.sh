set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
make -C oracle > /dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "prove or g2_fq or verify or padding or golden" > gpurun_out/pytest_logup.log 2>&1 || { tail -40 gpurun_out/pytest_logup.log; exit 1; }
tail -3 gpurun_out/pytest_logup.log
bash tools/gpu_lib_ab.sh libbn254stark_pre_logup.so
