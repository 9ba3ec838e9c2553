cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "gnt" 2>&1 | tail -3
bash tools/quick_trace.sh --config c4
head -8 gpurun_out/quick/steady.txt
cd $GRAFT_REPO_ROOT; python bench.py --config c4 --steps 5 --warmup 2 --extras 0 --cpu-iters 0 2>/dev/null | cut -c1-230
