#!/bin/bash
# round 5, call ap: blocks of 64 (matrix cores, fused arithmetic) against 32 between 0.5 and 1.1 GiB, and the smallest
# tableaus (blocks against two launches per pivot), same box
mkdir -p gpurun_out
O=gpurun_out/r05_block_policy_64_and_tiny.txt
: > $O
for shape in 8192x10240 8192x12288 6144x16384 8192x14336 12288x8192 16384x6144; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "block=32;block=64;block=32;block=64" 2048 64 >> $O 2>&1
done
for shape in 16x32 32x64 64x128 100x200; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "block=1;block=8;block=16;block=1;block=16" 1024 32 >> $O 2>&1
done
cat $O
