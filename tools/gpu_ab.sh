# A/B of two builds of the library (tuning only): A = libbn254stark.so, B = libvariantB.so
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
L=gpurun_out/ab/run.log
: > $L
cp plonky2_bn254_amd/libbn254stark.so /tmp/libA.so
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then cp /tmp/libA.so plonky2_bn254_amd/libbn254stark.so; else cp plonky2_bn254_amd/libvariantB.so plonky2_bn254_amd/libbn254stark.so; fi
    echo "== variant $v" | tee -a $L
    python tools/bench_hash.py 2>&1 | tee -a $L
    python tools/run_proofs.py 10 batch 2>&1 | tail -1 | tee -a $L
  done
done
cp /tmp/libA.so plonky2_bn254_amd/libbn254stark.so
