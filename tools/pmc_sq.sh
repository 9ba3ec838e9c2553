# Issue / stall counters (SQ block) of the kernels a program launches, one rocprofv3 --pmc pass per counter group.
# usage: bash tools/pmc_sq.sh <tag> <kernel-name substring> <python script> [args...]   -> gpurun_out/pmc_sq_<tag>.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FILTER=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o c -- python3 "$@" > $OUT/p$i.log 2>&1
done
FILTER="$FILTER" OUTDIR=$OUT python3 - <<'PY' > $REPO/gpurun_out/pmc_sq_$TAG.txt
import csv, glob, collections, os
out = os.environ['OUTDIR']
flt = os.environ['FILTER']
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        if flt not in k:
            continue
        tab[k][row['Counter_Name']].append(float(row['Counter_Value']))
def mean(cs, c):          # the median launch: a first launch that also sets up scratch / LDS limits does not skew the row
    v = sorted(cs.get(c, []))
    return v[len(v) // 2] if v else float('nan')
# GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_*_CYCLES of waves are in units of 4 clocks, SQ_VALU_MFMA_BUSY_CYCLES in clocks
print('%-44s %9s %10s %10s %10s %10s %9s' % ('kernel', 'launches', 'clocks', 'mfma busy', 'waves/SIMD', 'wait(cnt)', 'VALU:MFMA'))
for k, cs in sorted(tab.items(), key=lambda kv: -mean(kv[1], 'GRBM_GUI_ACTIVE') * len(kv[1].get('GRBM_GUI_ACTIVE', []))):
    clk = mean(cs, 'GRBM_GUI_ACTIVE') / 8.0
    print('%-44s %9d %10.0f %10.3f %10.2f %10.3f %9.1f' % (k[:44], len(cs.get('GRBM_GUI_ACTIVE', [])), clk,
          mean(cs, 'SQ_VALU_MFMA_BUSY_CYCLES') / (clk * 1024), 4 * mean(cs, 'SQ_WAVE_CYCLES') / (clk * 1024),
          mean(cs, 'SQ_WAIT_ANY') / mean(cs, 'SQ_WAVE_CYCLES'),
          (mean(cs, 'SQ_INSTS_VALU') - mean(cs, 'SQ_INSTS_MFMA')) / max(mean(cs, 'SQ_INSTS_MFMA'), 1.0)))
print()
for k, cs in tab.items():
    print(k)
    for c, v in sorted(cs.items()):
        print('   %-28s mean/launch %16.0f  (%d launches)' % (c, sum(v) / len(v), len(v)))
PY
cat $REPO/gpurun_out/pmc_sq_$TAG.txt
rm -rf $OUT
