# Round profile on the GPU box: bench lines, rocprofv3 kernel trace + stats of the timed steps, PMC traffic passes (each counter
# in its own run), steady-state kernel table and step timeline -- for BASELINE config 2 (headline) and, with trace + PMC as well,
# configs 4 and 5.  Writes gpurun_out/final; copy what is to be judged into profiles/.
set -x
ROUND=${ROUND:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench_ibrnet.json 2> $OUT/bench_ibrnet.err
python3 $REPO/bench.py --steps 1000 --warmup 3 --extras 0 --cpu-iters 0 > $OUT/bench_1000iters_ibrnet.json 2> $OUT/bench_1000.err
profile() {   # $1 tag, $2.. bench flags
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o t -- python3 $REPO/bench.py "$@" --steps 10 --warmup 3 --extras 0 --event-every 0 > $OUT/trace_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcf_$tag -o f -- python3 $REPO/bench.py "$@" --steps 3 --warmup 2 --extras 0 --event-every 0 > $OUT/pmcf_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcw_$tag -o w -- python3 $REPO/bench.py "$@" --steps 3 --warmup 2 --extras 0 --event-every 0 > $OUT/pmcw_$tag.log 2>&1
  T=$(ls $OUT/trace_$tag/*/*kernel_trace.csv $OUT/trace_$tag/*kernel_trace.csv 2>/dev/null | head -1)
  F=$(ls $OUT/pmcf_$tag/*/*counter_collection.csv $OUT/pmcf_$tag/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $OUT/pmcw_$tag/*/*counter_collection.csv $OUT/pmcw_$tag/*counter_collection.csv 2>/dev/null | head -1)
  python3 $REPO/tools/steady_state_kernels.py $T 10 60 > $OUT/steady_state_kernels_$tag.txt
  python3 $REPO/tools/step_timeline.py $T > $OUT/step_timeline_$tag.txt 2>&1
  cp $(ls $OUT/trace_$tag/*/*kernel_stats.csv $OUT/trace_$tag/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_$tag.csv
  python3 $REPO/tools/pmc_kernels.py $F $W 3 > $OUT/pmc_traffic_$tag.txt 2>&1
  if [ "$tag" = c2 ]; then python3 $REPO/tools/pmc_traffic.py $F $W 3 $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; fi
  rm -rf $OUT/trace_$tag $OUT/pmcf_$tag $OUT/pmcw_$tag
}
profile c2
profile c4 --config c4
profile c5 --config c5
# issue / stall counters of the same steps (matrix-pipe busy fraction, waves per SIMD, s_waitcnt share, vector : matrix instructions)
sq() { tag=$1; shift; bash $REPO/tools/pmc_sq.sh $tag k_ $REPO/bench.py "$@" --steps 3 --warmup 2 --extras 0 --event-every 0 --cpu-iters 0 > /dev/null 2>&1; head -34 $REPO/gpurun_out/pmc_sq_$tag.txt > $OUT/sq_pipe_$tag.txt; }
sq c2
sq c4 --config c4
sq c5 --config c5
cd /tmp
python3 $REPO/bench.py --config c4 --steps 5 --warmup 2 --render-chunks 4 > $OUT/bench_gnt.json 2> $OUT/bench_gnt.err
python3 $REPO/bench.py --config c5 --steps 10 --warmup 3 > $OUT/bench_c5_bf16.json 2> $OUT/bench_c5.err
python3 $REPO/bench.py --config c5 --precision fp32 --steps 10 --warmup 3 > $OUT/bench_c5_fp32.json 2> $OUT/bench_c5_fp32.err
bash $REPO/tools/pmc_render.sh > $OUT/pmc_render.log 2>&1; cp $REPO/gpurun_out/pmc_render.txt $OUT/pmc_render_traffic.txt
cd $REPO && timeout 900 python3 -m pytest tests -m gpu -q -s 2>&1 | grep -E "grad parity|config 5|full size|bf16 attack|passed|failed" > $OUT/parity_numbers.txt
ls -la $OUT
