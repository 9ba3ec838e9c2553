# gpurun_out/final (written by tools/profile_round.sh on the GPU box) -> profiles/rNN_*, then the summary
ROUND=${1:-r05}
cd "$(dirname "$0")/.."
F=gpurun_out/final; P=profiles
cp $F/bench_ibrnet.json $P/${ROUND}_bench_ibrnet.json; cp $F/bench_1000iters_ibrnet.json $P/${ROUND}_bench_1000iters_ibrnet.json
cp $F/bench_gnt.json $P/${ROUND}_bench_gnt.json; cp $F/bench_c5_bf16.json $P/${ROUND}_bench_c5_bf16.json; cp $F/bench_c5_fp32.json $P/${ROUND}_bench_c5_fp32.json
cp $F/pmc_traffic.json $P/${ROUND}_pmc_traffic.json; cp $F/pmc_render_traffic.txt $P/${ROUND}_pmc_render_traffic.txt; cp $F/parity_numbers.txt $P/${ROUND}_parity_numbers.txt
for c in c2 c4 c5; do
  cp $F/steady_state_kernels_$c.txt $P/${ROUND}_steady_state_kernels_$c.txt; cp $F/step_timeline_$c.txt $P/${ROUND}_step_timeline_$c.txt
  cp $F/pmc_traffic_$c.txt $P/${ROUND}_pmc_traffic_$c.txt; cp $F/rocprofv3_kernel_stats_$c.csv $P/${ROUND}_rocprofv3_kernel_stats_$c.csv
  cp $F/sq_pipe_$c.txt $P/${ROUND}_sq_pipe_$c.txt
done
python tools/make_profile_summary.py $ROUND
