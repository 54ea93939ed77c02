#!/bin/bash
# round 4, call ax: which job of k_block_fixup takes its time (diagnostic builds that leave one job out; results wrong)
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in fixskip0 fixskip1; do
  export LPX_LIB_PATH=$R/gpurun_variants/liblpx_$L.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_ax_$L -- python3 $R/scripts/arith_grid.py cfg4 "fused=1" 320 64 > $R/gpurun_out/r04_ax.log 2>&1
  find $R/gpurun_out/r04_ax_$L -name "*kernel_trace.csv" -delete; find $R/gpurun_out/r04_ax_$L -name "*agent_info.csv" -delete
done
cd $R
python - <<'PY'
import csv,glob
for d in sorted(glob.glob('gpurun_out/r04_ax_*')):
    f=glob.glob(d+'/**/*kernel_stats.csv', recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        if 'k_block_fixup' in r['Name'] and int(r['Calls'])>3:
            print(d.split('r04_ax_')[1], r['Calls'], "k_block_fixup avg us %.1f max %.1f" % (float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
