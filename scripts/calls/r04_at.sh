#!/bin/bash
# round 4, call at: by-size block policy below 1 GiB, both arithmetic modes: blocks of 16 / 32 (and by size)
mkdir -p gpurun_out
O=gpurun_out/r04_at.txt
: > $O
for S in 4096x8192 6144x8192 8192x8192 4096x16384 6144x16384; do
  for F in 0 1; do
    timeout -k 10 200 python scripts/arith_grid.py $S "fused=$F;fused=$F,block=16;fused=$F,block=32" 512 64 2>&1 | grep -v "^#" | sed "s/^/$S /" >> $O
  done
done
cat $O
