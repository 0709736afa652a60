# single-proof stage times of all three kinds: shipped library against tools/ubench/ab/$1
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/tools/ubench/ab/$1
for rep in 1 2; do
echo -n "shipped: "; python tools/run_proofs.py 6 single | head -1
echo -n "other:   "; env BN254S_LIB=$OTHER python tools/run_proofs.py 6 single | head -1
done
