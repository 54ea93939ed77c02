#!/bin/bash
# round 4, call av: k_block_fixup with the uniform tests hoisted (one mask per pending pivot and workgroup, one ballot per
# wave): parity, then its duration under rocprofv3 (cfg4 fused / default, cfg3) and the grids
R=$PWD
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -k "blocked or ragged or degenerate or restart or cycling or cfg5 or shard or multi_loop or 16_row or in_place or wide_decision or by_size or golden" > gpurun_out/r04_av_quick.log 2>&1
tail -3 gpurun_out/r04_av_quick.log
cd /tmp && export TMPDIR=/tmp
for W in "cfg4 fused=1" "cfg4 fused=0" "cfg3 fused=1"; do
  set -- $W
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_av_$1_$2 -- python3 $R/scripts/arith_grid.py $1 "$2" 320 64 > $R/gpurun_out/r04_av.log 2>&1
  echo "== $1 $2: $(grep pivots/s $R/gpurun_out/r04_av.log | cut -c1-120)"
  find $R/gpurun_out/r04_av_$1_$2 -name "*kernel_trace.csv" -delete; find $R/gpurun_out/r04_av_$1_$2 -name "*agent_info.csv" -delete
done
cd $R
python - <<'PY'
import csv,glob
for d in sorted(glob.glob('gpurun_out/r04_av_cfg*')):
    f=glob.glob(d+'/**/*kernel_stats.csv', recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        if 'k_block_fixup' in r['Name'] and int(r['Calls'])>3:
            print(d.split('r04_av_')[1], r['Calls'], "k_block_fixup avg us %.1f max %.1f" % (float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1" 512 64 2>&1 | grep -v "^#"
timeout -k 10 300 python scripts/arith_grid.py 12288x16384 "fused=0;fused=1" 512 64 2>&1 | grep -v "^#"
