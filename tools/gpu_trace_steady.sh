# rocprofv3 kernel trace of a longer bench run and where it leaves the GPU without a wide kernel (tools/trace_gaps.py); tuning only.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_steady
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o steady --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/steady_kernel_trace.csv > gpurun_out/trace_gaps_steady.txt 2>&1
cat gpurun_out/trace_gaps_steady.txt
rm -f $O/steady_kernel_trace.csv
