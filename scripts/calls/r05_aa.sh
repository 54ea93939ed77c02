#!/bin/bash
# round 5, call aa: where the driver's 20-pivot command spends its time (serial one-block form at cfg4): kernel-trace timeline
mkdir -p gpurun_out
R=$PWD
O=$R/gpurun_out/r05_aa_timeline.txt
: > $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05_aa_trace -- python3 $R/scripts/one_block_call.py cfg4 20 5 4 > $R/gpurun_out/r05_aa.log 2>&1
grep "call\|block" $R/gpurun_out/r05_aa.log >> $O
T=$(find $R/gpurun_out/r05_aa_trace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/trace_timeline.py $T 0.6 60 >> $O 2>&1
cd $R
timeout -k 10 200 python scripts/one_block_call.py cfg4 20 5 6 >> $O 2>&1
cat $O
