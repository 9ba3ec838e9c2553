# Round profile on the GPU box: bench lines, rocprofv3 kernel trace + stats of the timed steps, PMC traffic passes (each counter
# in its own run), steady-state kernel table.  Writes gpurun_out/final; copy what is to be judged into profiles/.
set -x
ROUND=${ROUND:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench_ibrnet.json 2> $OUT/bench_ibrnet.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $ROUND -- python3 $REPO/bench.py --steps 10 --warmup 3 --extras 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $REPO/bench.py --steps 3 --warmup 2 --extras 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $REPO/bench.py --steps 3 --warmup 2 --extras 0 > $OUT/pmc_write.log 2>&1
python3 $REPO/tools/steady_state_kernels.py $(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1) 10 60 > $OUT/steady_state_kernels.txt
python3 $REPO/tools/pmc_traffic.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv $OUT/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv $OUT/pmc_write/*counter_collection.csv 2>/dev/null | head -1) 3 $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv
python3 $REPO/bench.py --config c4 --steps 5 --warmup 2 --render-chunks 2 > $OUT/bench_gnt.json 2> $OUT/bench_gnt.err
python3 $REPO/bench.py --config c5 --steps 10 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv $OUT/pmc_fetch/*/*.csv $OUT/pmc_write/*/*.csv $OUT/pmc_fetch/*.csv $OUT/pmc_write/*.csv
ls -la $OUT
