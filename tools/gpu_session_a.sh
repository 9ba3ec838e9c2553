# Round-3 GPU session A: parity suite, the cost of the bench's own HIP-event brackets, the 2-rank functional run with the sharded
# render legs, and rocprofv3 evidence for configs 4 and 5 (kernel trace + steady-state table + PMC traffic, one counter per run).
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
timeout 1800 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for e in 1 4 0; do
  timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-iters 0 --event-every $e > $OUT/bench_events_$e.json 2> $OUT/bench_events_$e.err
done
NERFOOL_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 4 --warmup 1 --cpu-iters 0 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
tail -3 $OUT/bench_2rank_gloo.err
cd /tmp && export TMPDIR=/tmp
for cfg in c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -o t -- python3 $REPO/bench.py --config $cfg --steps 6 --warmup 3 --extras 0 --event-every 0 > $OUT/trace_$cfg.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcf_$cfg -o f -- python3 $REPO/bench.py --config $cfg --steps 3 --warmup 2 --extras 0 --event-every 0 > $OUT/pmcf_$cfg.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcw_$cfg -o w -- python3 $REPO/bench.py --config $cfg --steps 3 --warmup 2 --extras 0 --event-every 0 > $OUT/pmcw_$cfg.log 2>&1
  T=$(ls $OUT/trace_$cfg/*/*kernel_trace.csv $OUT/trace_$cfg/*kernel_trace.csv 2>/dev/null | head -1)
  python3 $REPO/tools/steady_state_kernels.py $T 6 60 > $OUT/steady_state_$cfg.txt
  python3 $REPO/tools/step_timeline.py $T > $OUT/step_timeline_$cfg.txt 2>&1
  cp $(ls $OUT/trace_$cfg/*/*kernel_stats.csv $OUT/trace_$cfg/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats_$cfg.csv
  python3 $REPO/tools/pmc_kernels.py $(ls $OUT/pmcf_$cfg/*/*counter_collection.csv $OUT/pmcf_$cfg/*counter_collection.csv 2>/dev/null | head -1) \
     $(ls $OUT/pmcw_$cfg/*/*counter_collection.csv $OUT/pmcw_$cfg/*counter_collection.csv 2>/dev/null | head -1) 3 > $OUT/pmc_kernels_$cfg.txt 2>&1
  rm -rf $OUT/trace_$cfg $OUT/pmcf_$cfg $OUT/pmcw_$cfg
done
cd $REPO
timeout 600 python bench.py --config c4 --steps 5 --warmup 2 --render-chunks 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python bench.py --config c5 --precision fp32 --steps 10 --warmup 3 > $OUT/bench_c5_fp32.json 2> $OUT/bench_c5_fp32.err
ls -la $OUT
