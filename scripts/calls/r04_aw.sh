#!/bin/bash
# round 4, call aw: k_block_fixup with its ring values requested one batch ahead: parity subset, duration, grids
R=$PWD
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "blocked or ragged or degenerate or restart or cycling or cfg5 or 16_row or in_place" > gpurun_out/r04_aw_quick.log 2>&1
tail -2 gpurun_out/r04_aw_quick.log
cd /tmp && export TMPDIR=/tmp
for W in "cfg4 fused=1" "cfg4 fused=0" "cfg3 fused=1"; do
  set -- $W
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_aw_$1_$2 -- python3 $R/scripts/arith_grid.py $1 "$2" 320 64 > $R/gpurun_out/r04_aw.log 2>&1
  find $R/gpurun_out/r04_aw_$1_$2 -name "*kernel_trace.csv" -delete; find $R/gpurun_out/r04_aw_$1_$2 -name "*agent_info.csv" -delete
done
cd $R
python - <<'PY'
import csv,glob
for d in sorted(glob.glob('gpurun_out/r04_aw_cfg*')):
    f=glob.glob(d+'/**/*kernel_stats.csv', recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        if 'k_block_fixup' in r['Name'] and int(r['Calls'])>3:
            print(d.split('r04_aw_')[1], r['Calls'], "k_block_fixup avg us %.1f max %.1f" % (float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1" 512 64 2>&1 | grep -v "^#"
