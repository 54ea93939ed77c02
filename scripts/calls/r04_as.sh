#!/bin/bash
# round 4, call as: the by-size block policy of the fused mode after lowering the threshold of blocks of 64 to ~1.2 GiB
mkdir -p gpurun_out
O=gpurun_out/r04_as.txt
: > $O
for S in 9216x16384 10240x16384 11264x16384 12288x16384 14336x16384 8192x16384; do
  timeout -k 10 200 python scripts/arith_grid.py $S "fused=1;fused=1,block=32;fused=1,block=64" 512 64 2>&1 | grep -v "^#" | sed "s/^/$S /" >> $O
done
cat $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "fused and (cfg3 or cfg4 or beyond)" 2>&1 | tail -2
