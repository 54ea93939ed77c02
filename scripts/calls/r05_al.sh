#!/bin/bash
# round 5, call al: the loop's events as stop events of the launches they follow (hipExtLaunchKernelGGL) instead of records behind
# them: the launch-gap micro benchmark, parity (blocked tests, shards), then cfg3 / cfg4 / two mid sizes against the previous
# commit's library (gpurun_variants/liblpx_prev.so), same box
mkdir -p gpurun_out
timeout -k 10 120 scripts/micro/launch_gap > gpurun_out/r05_launch_gap.txt 2>&1
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "block or 64 or cfg4 or cfg3 or by_size or ladders or timed or fixup or multi or shard or golden or degenerate" > gpurun_out/r05_al_gpu.log 2>&1
tail -3 gpurun_out/r05_al_gpu.log
O=gpurun_out/r05_stop_events_ab.txt
: > $O
for rep in 1 2 3; do
  for cfg in cfg3 cfg4 4096x8192 2048x4096; do
    echo "## prev $cfg" >> $O
    LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so timeout -k 10 120 python scripts/arith_grid.py $cfg "block=0" 1024 64 2>&1 | grep pivots/s >> $O
    echo "## new $cfg" >> $O
    timeout -k 10 120 python scripts/arith_grid.py $cfg "block=0" 1024 64 2>&1 | grep pivots/s >> $O
  done
done
cat $O
