set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -s -k "bf16 or render_rays or stage" 2>&1 | grep -E "config 5|bf16 attack|passed|failed|Error" | tail -20
for i in 1 2; do
timeout 600 python bench.py --config c5 --steps 10 --warmup 3 --render-chunks 0 > $OUT/bench_c5_$i.json 2> $OUT/err
timeout 600 python bench.py --config c5 --precision fp32 --steps 10 --warmup 3 --render-chunks 0 > $OUT/bench_c5_fp32_$i.json 2> $OUT/err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r03e/bench_c5*.json')):
    p=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(p['ms_per_step'],3), {k:(v['mean_ms'],v['frac']) for k,v in p['extra']['kernels'].items()})
PY
python tools/bench_ibrnet_kernels.py 20 2>&1 | grep bf16
