# A/B on one box: build/variants/base.so (before the DPP reductions) against the current library
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
python tools/bench_ibrnet_kernels.py 20 build/variants/base.so > $OUT/ibr_base.txt 2>&1
python tools/bench_ibrnet_kernels.py 20 > $OUT/ibr_new.txt 2>&1
python tools/bench_gnt_kernels.py 5 build/variants/base.so > $OUT/gnt_base.txt 2>&1
python tools/bench_gnt_kernels.py 5 > $OUT/gnt_new.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err
paste $OUT/ibr_base.txt $OUT/ibr_new.txt | cut -c1-250
cat $OUT/gnt_base.txt $OUT/gnt_new.txt
