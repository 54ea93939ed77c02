#!/bin/bash
# round 5, call ao: the block policy at the small end (two launches per pivot below ~18 MiB was round 3's measurement) and
# blocks of 64 in the decision-bound range, with the round-5 decision kernel, same box
mkdir -p gpurun_out
O=gpurun_out/r05_block_policy_small_sizes.txt
: > $O
for shape in 128x256 256x512 512x512 512x1024 1024x1024 512x4096 1024x2048 1536x2048; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "block=1;block=8;block=16;block=32;block=1;block=16" 2048 64 >> $O 2>&1
done
for shape in 4096x8192 8192x8192 cfg3; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "block=32;block=64;block=32;block=64" 2048 64 >> $O 2>&1
done
cat $O
