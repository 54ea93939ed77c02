#!/bin/bash
# round 5, call ag: kernel-trace timeline of the steady loop on the final tree (fix-up chains and pack kernel on the third
# stream): what is left between two sweeps on the sweep stream, cfg4 and cfg3
mkdir -p gpurun_out
R=$PWD
O=$R/gpurun_out/r05_ag_timeline.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for cfg in cfg4 cfg3; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05_ag_trace_$cfg -- python3 $R/scripts/arith_grid.py $cfg "block=0" 384 64 > $R/gpurun_out/r05_ag_$cfg.log 2>&1
  T=$(find $R/gpurun_out/r05_ag_trace_$cfg -name "*kernel_trace.csv" | head -1)
  echo "## $cfg, default options" >> $O
  grep "pivots/s" $R/gpurun_out/r05_ag_$cfg.log >> $O
  python3 $R/scripts/trace_timeline.py $T 0.8 24 >> $O 2>&1
done
cat $O
