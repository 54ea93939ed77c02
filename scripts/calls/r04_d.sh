#!/bin/bash
# round 4, call d: where a decision of k_block_chain2 spends its time (eight stamps), alone and beside the sweep
mkdir -p gpurun_out
O=gpurun_out/r04_d_trace.txt
: > $O
for W in cfg3 cfg4; do
  for X in "overlap=0" "" "fused=1" "fused=1 chain_cus=8"; do
    timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -3 >> $O
  done
done
cat $O
