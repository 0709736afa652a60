# Round profiles (run on the GPU box): rocprofv3 kernel stats of the bench command, SQ_INSTS_VALU per kernel, HBM traffic of the
# NTT/LDE stage (separate --pmc passes), the leaf-hash kernel alone, and the instruction-issue microbenchmarks.
# usage: bash tools/gpu_profiles.sh rNN   -> gpurun_out/profiles_rNN/ (copy what is to be judged into profiles/)
set -e
R=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles_$R
mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_prof.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O -o insts --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_proofs.py 2 single > $O/insts.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O -o hash --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_hash.py > $O/hash_pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O -o fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_ntt.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O -o write --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_ntt.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O -o ntt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pmc_ntt.py > $O/ntt_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/bench_hash.py sweep > $O/hash_sweep.log 2>&1
timeout -k 5 60 tools/ubench/sgpr_ops 2>&1 | tr "\r" "\n" | grep -v "\.\.\.$" > $O/ubench_issue.log || true
timeout -k 5 100 tools/ubench/poseidon_ub > $O/ubench_poseidon.log 2>&1 || true
ls $O
