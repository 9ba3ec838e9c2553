# HBM traffic of the render path's kernels, gather fused vs separate (one counter per run); writes gpurun_out/pmc_render.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_render
mkdir -p $OUT
for mode in fused separate; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${mode}_$c -o p -- python3 $REPO/tools/render_chunks.py 8 $mode > /dev/null 2>&1
  done
done
python3 - <<'PY' > $REPO/gpurun_out/pmc_render.txt
import csv, glob, collections, os
out = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc_render'
for mode in ('fused', 'separate'):
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for ci, c in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
        f = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (out, mode, c), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if name.startswith(('k_ibr_rows_fwd', 'k_ibr_ray_fwd', 'k_project_gather_fwd')):
                tot[name.split('<')[0]][ci] += float(r['Counter_Value'])
                tot[name.split('<')[0]][2] += 1 if ci == 0 else 0
    print(mode)
    s = 0.0
    for k, (f, w, n) in sorted(tot.items()):
        b = (2 * f + w) * 1024 / max(n, 1)
        s += b
        print('  %-24s %3d launches  %.1f MB per launch (2 x FETCH + WRITE)' % (k, n, b / 1e6))
    print('  sum per 4096-ray chunk: %.1f MB' % (s / 1e6))
PY
rm -rf $OUT
cat $REPO/gpurun_out/pmc_render.txt
