REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/quick
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py "$@" --steps 10 --warmup 3 --extras 0 --event-every 0 > $OUT/trace.log 2>&1
T=$(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1)
python3 $REPO/tools/steady_state_kernels.py $T 10 60 > $OUT/steady.txt
rm -rf $OUT/trace
