set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python tools/bench_ibrnet_kernels.py 20 > $OUT/ibr_new.txt 2>&1; cat $OUT/ibr_new.txt
timeout 600 python bench.py --steps 20 --warmup 3 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; p=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(p['ms_per_step'], p['value']); print({k:(v['mean_ms'],v['frac']) for k,v in p['extra']['kernels'].items()}); print(p['extra']['render_800x800_64'])"
