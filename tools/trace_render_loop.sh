cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/rt -o t -- python3 $REPO/tools/render_chunks.py 24 fused > /dev/null 2>&1
T=$(ls $REPO/gpurun_out/rt/*/*kernel_trace.csv $REPO/gpurun_out/rt/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$T" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 16 chunks: find k_ibr_rows_fwd launches
idx=[i for i,r in enumerate(rows) if 'k_ibr_rows_fwd' in r['Kernel_Name']]
lo,hi=idx[-17],idx[-1]
t0=int(rows[lo]['Start_Timestamp']); t1=int(rows[hi]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[lo:hi])
print('16 chunks: window %.3f ms, busy %.3f ms, per chunk %.1f us window / %.1f us busy' % ((t1-t0)/1e6, busy/1e6, (t1-t0)/16e3, busy/16e3))
prev=None
for r in rows[idx[-2]:idx[-1]+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((s-int(rows[idx[-2]]['Start_Timestamp']))/1e3,(e-s)/1e3, 0 if prev is None else (s-prev)/1e3, r['Kernel_Name'].split('(')[0][:50]))
    prev=e
PY
rm -rf $REPO/gpurun_out/rt
