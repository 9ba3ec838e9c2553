cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "attack or scatter or ibrnet or c5 or grad" 2>&1 | tail -3
bash tools/quick_trace.sh --config c5
grep -E "window|rows_bwd|ray_bwd" gpurun_out/quick/steady.txt
cd $GRAFT_REPO_ROOT; bash tools/quick_trace.sh
grep -E "window|rows_bwd|ray_bwd" gpurun_out/quick/steady.txt
cd $GRAFT_REPO_ROOT; python bench.py --config c5 --steps 10 --warmup 3 --extras 0 --cpu-iters 0 2>/dev/null | cut -c1-230
python bench.py --steps 20 --warmup 3 --extras 0 --cpu-iters 0 2>/dev/null | cut -c1-230
