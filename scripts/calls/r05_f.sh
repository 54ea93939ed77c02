#!/bin/bash
# round 5, call f: kernel-trace timeline of the default loop at cfg4 (by-size arithmetic = fused, blocks of 64 on the matrix
# cores): what sits between two sweeps on the sweep stream ("other" of loop_bound), per-kernel durations and gaps
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_f_trace -- python3 $R/scripts/arith_grid.py cfg4 "block=0" 384 64 > $R/gpurun_out/r05_f.log 2>&1
cd $R
tail -2 gpurun_out/r05_f.log
O=gpurun_out/r05_f_timeline.txt
: > $O
F=$(find gpurun_out/r05_f_trace -name "*kernel_stats.csv" | head -1)
head -12 $F | cut -c1-170 >> $O
T=$(find gpurun_out/r05_f_trace -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T >> $O 2>&1
python scripts/trace_timeline.py $T 0.7 28 >> $O 2>&1
cat $O
