#!/bin/bash
# round 5, call ak: the clock probe's front stamp moves from the pack kernel (now on another stream, long before the sweep) to the
# start of the sweep kernel itself: parity of the blocked tests + variants, the probe by placement, throughput
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu -k "block or 64 or cfg4 or cfg3 or by_size or ladders or timed or fixup or variants or superseded or round3 or first_mfma or pair_of_waves" > gpurun_out/r05_ak_gpu.log 2>&1
tail -3 gpurun_out/r05_ak_gpu.log
O=gpurun_out/r05_clock_probe_by_placement.txt
: > $O
timeout -k 10 120 python scripts/clock_probe_check.py cfg3 8 >> $O 2>&1
timeout -k 10 120 python scripts/clock_probe_check.py cfg4 6 >> $O 2>&1
for rep in 1 2; do
  timeout -k 10 120 python scripts/arith_grid.py cfg4 "block=0" 1024 64 2>&1 | grep pivots/s >> $O
  timeout -k 10 120 python scripts/arith_grid.py cfg3 "block=0;fused=0" 1024 64 2>&1 | grep pivots/s >> $O
done
cat $O
