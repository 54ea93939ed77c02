#!/bin/bash
# round 4, call ar: fused mode, blocks of 32 against blocks of 64 (k_sweep64_mfma2) by tableau size, same box, twice
mkdir -p gpurun_out
O=gpurun_out/r04_ar.txt
: > $O
for k in 1 2; do
  for S in 8192x16384 8192x8192 12288x16384 4096x16384 16384x8192; do
    timeout -k 10 200 python scripts/arith_grid.py $S "fused=1,block=32;fused=1,block=64" 512 64 2>&1 | grep -v "^#" | sed "s/^/$S /" >> $O
  done
done
cat $O
