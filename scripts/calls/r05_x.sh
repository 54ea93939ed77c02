#!/bin/bash
# round 5, call x: kernel-trace timelines of the steady loop at cfg4 with the fix-up behind the sweep (LPX_OPT_FIXUP_SIDE 0) and
# beside it on the decisions' CUs (2) / the sweep's CUs (1): what is left between two sweeps on the sweep stream
mkdir -p gpurun_out
R=$PWD
O=$R/gpurun_out/r05_x_timeline.txt
: > $O
cd /tmp && export TMPDIR=/tmp
for mode in 0 2 1; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05_x_trace$mode -- python3 $R/scripts/arith_grid.py cfg4 "fixup_side=$mode" 384 64 > $R/gpurun_out/r05_x$mode.log 2>&1
  tail -1 $R/gpurun_out/r05_x$mode.log
  T=$(find $R/gpurun_out/r05_x_trace$mode -name "*kernel_trace.csv" | head -1)
  echo "## LPX_OPT_FIXUP_SIDE = $mode" >> $O
  grep "pivots/s" $R/gpurun_out/r05_x$mode.log >> $O
  python3 $R/scripts/trace_timeline.py $T 0.75 26 >> $O 2>&1
done
cat $O
