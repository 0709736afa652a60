# A/B of two builds of the library on one box: the shipped one against tools/ubench/ab/$1 (git-ignored), alternating.
# usage: bash tools/gpu_lib_ab.sh libbn254stark_xxx.so
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/tools/ubench/ab/$1
run() {
  echo -n "$1 : "
  env $2 python bench.py --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_kinds']; print(d['value'], 'seq', d['sequential_steps']['value'], 'g2', o['g2_scalar_mul']['proofs_per_s'], 'fq', o['fq_exp']['proofs_per_s'], 'tall19', d['one_tall_proof_of_1024']['ms_per_proof'])"
}
echo -n "single proofs, shipped: "; python tools/run_proofs.py 4 single | head -1
echo -n "single proofs, other:   "; env BN254S_LIB=$OTHER python tools/run_proofs.py 4 single | head -1
run shipped A=1
run other BN254S_LIB=$OTHER
run shipped A=2
run other BN254S_LIB=$OTHER
