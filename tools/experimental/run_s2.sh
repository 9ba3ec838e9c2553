cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "conv_s2" 2>&1 | tail -2
for v in "" build/variants/st81.so build/variants/st82.so; do echo "== $v"; python tools/bench_conv_s2.py $v 2>&1 | grep "stem forward"; done | tee gpurun_out/s2_stem.txt
