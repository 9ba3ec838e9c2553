set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
./tools/experimental/probe_overlap > $OUT/probe_overlap.txt 2>&1
cat $OUT/probe_overlap.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python tools/bench_ibrnet_kernels.py 20 > $OUT/ibr_new.txt 2>&1
python tools/bench_gnt_kernels.py 5 > $OUT/gnt_new.txt 2>&1
cat $OUT/ibr_new.txt $OUT/gnt_new.txt
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err
