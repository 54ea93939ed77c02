#!/bin/bash
# round 5, call w: the fix-up's chains beside the sweep (LPX_OPT_FIXUP_SIDE): parity of the blocked tests with the new default,
# then throughput by mode (0 = behind the sweep as before; 1 sweep's CUs; 2 decisions' CUs; 3 no mask), cfg4 and cfg3, both arithmetics
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block or 64 or cfg4 or cfg3 or by_size or ladders or beyond or timed" > gpurun_out/r05_w_gpu.log 2>&1
tail -3 gpurun_out/r05_w_gpu.log
O=gpurun_out/r05_w_ab.txt
: > $O
M="fixup_side=0;fixup_side=1;fixup_side=2;fixup_side=3;fixup_side=0;fixup_side=1;fixup_side=2;fixup_side=3"
P="fused=0,fixup_side=0;fused=0,fixup_side=1;fused=0,fixup_side=2;fused=0,fixup_side=0;fused=0,fixup_side=1;fused=0,fixup_side=2"
for cfg in cfg4 cfg3; do
  timeout -k 10 300 python scripts/arith_grid.py $cfg "$M" 1024 64 >> $O 2>&1
  timeout -k 10 300 python scripts/arith_grid.py $cfg "$P" 1024 64 >> $O 2>&1
done
cat $O
