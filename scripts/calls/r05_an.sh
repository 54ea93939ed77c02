#!/bin/bash
# round 5, call an: the by-size block policy below 330 MiB re-measured with the round-5 decision kernel and launches (a block
# costs ~17 us besides its decisions: ~12 us from the end of one decision launch to the entry of the next, 4-6 us of prologue —
# profiles/r05_chain_launch_stamps.txt): blocks of 16 (the policy) against 32 and 24, same box
mkdir -p gpurun_out
O=gpurun_out/r05_block_policy_mid_sizes.txt
: > $O
for shape in 1024x2048 2048x2048 2048x4096 4096x4096 4096x8192 6144x6144 2048x16384; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "block=0;block=32;block=24;block=16;block=32" 2048 64 >> $O 2>&1
done
cat $O
