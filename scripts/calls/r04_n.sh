#!/bin/bash
# round 4, call n: kernel trace of the fused loop with blocks of 64 at cfg4: what sits between two sweeps
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_n_trace -- python3 $R/scripts/arith_grid.py cfg4 "fused=1,block=64" 320 64 > $R/gpurun_out/r04_n.log 2>&1
cd $R
tail -3 gpurun_out/r04_n.log
F=$(find gpurun_out/r04_n_trace -name "*kernel_stats.csv" | head -1)
head -14 $F | cut -c1-160
T=$(find gpurun_out/r04_n_trace -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T 2>&1 | tail -30
