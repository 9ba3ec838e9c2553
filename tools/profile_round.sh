set -x
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
python3 /root/repo/bench.py --steps 20 --warmup 3 > $OUT/bench_ibrnet.json 2> $OUT/bench_ibrnet.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r01 -- python3 /root/repo/bench.py --steps 10 --warmup 3 --cpu-iters 0 --render-chunks 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 /root/repo/bench.py --steps 3 --warmup 2 --cpu-iters 0 --render-chunks 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 /root/repo/bench.py --steps 3 --warmup 2 --cpu-iters 0 --render-chunks 0 > $OUT/pmc_write.log 2>&1
python3 /root/repo/bench.py --model gnt --steps 5 --warmup 2 --render-chunks 2 > $OUT/bench_gnt.json 2> $OUT/bench_gnt.err
rm -f $OUT/trace/*kernel_trace.csv.bak
ls -la $OUT $OUT/trace $OUT/pmc_fetch
