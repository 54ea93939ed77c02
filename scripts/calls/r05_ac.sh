#!/bin/bash
# round 5, call ac: the serial one-block form at cfg4 (the driver's command) by decision grid: 64 (default), 96, 128 workgroups
mkdir -p gpurun_out
O=gpurun_out/r05_ac_one_block_grid.txt
: > $O
for w in 64 128 96 64 128; do
  echo "## chain_wgs=$w" >> $O
  timeout -k 10 200 python scripts/one_block_call.py cfg4 20 5 6 chain_wgs=$w >> $O 2>&1
done
cat $O
