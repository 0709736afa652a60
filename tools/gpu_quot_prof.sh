# Per-kernel time and VALU instruction counts of single proofs, shipped library against tools/ubench/ab/$1
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/quot_prof
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o new_t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_proofs.py 4 single > $O/new_t.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O -o new_i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_proofs.py 2 single > $O/new_i.log 2>&1
export BN254S_LIB=$GRAFT_REPO_ROOT/tools/ubench/ab/$1
rocprofv3 --kernel-trace --stats -d $O -o old_t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_proofs.py 4 single > $O/old_t.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O -o old_i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_proofs.py 2 single > $O/old_i.log 2>&1
ls $O
