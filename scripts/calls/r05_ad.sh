#!/bin/bash
# round 5, call ad: the sweep's pack kernel on the side stream too (multipliers of block k packed while sweep k-1 still runs;
# packed-multiplier buffer and ticket counters once per ring half): parity of the blocked tests, then cfg3 / cfg4 against the
# previous commit's library (gpurun_variants/liblpx_prev.so), same box
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block or 64 or cfg4 or cfg3 or by_size or ladders or beyond or timed or fixup" > gpurun_out/r05_ad_gpu.log 2>&1
tail -3 gpurun_out/r05_ad_gpu.log
O=gpurun_out/r05_ad_ab.txt
: > $O
for rep in 1 2 3; do
  for cfg in cfg3 cfg4; do
    echo "## prev $cfg" >> $O
    LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so timeout -k 10 120 python scripts/arith_grid.py $cfg "block=0;fused=0" 1024 64 >> $O 2>&1
    echo "## new $cfg" >> $O
    timeout -k 10 120 python scripts/arith_grid.py $cfg "block=0;fused=0" 1024 64 >> $O 2>&1
  done
done
grep -v "^# " $O
