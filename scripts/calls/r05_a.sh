#!/bin/bash
# round 5, call a: (1) per-instruction account of a pending-pivot chain step on a lone wave (scripts/micro/chain_step_cost.hip);
# (2) the "before" 16-stamp traces of k_block_chain2 at HEAD of round 4 on this box (fine build) + 8-stamp release traces
mkdir -p gpurun_out
O=gpurun_out/r05_a.txt
: > $O
timeout -k 10 120 scripts/micro/chain_step_cost >> $O 2>&1
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
for W in cfg3 cfg4; do
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 fused=1 2>&1 | tail -2 >> $O
done
unset LPX_LIB_PATH
for W in cfg3 cfg4; do
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 2>&1 | tail -3 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 fused=1 2>&1 | tail -3 >> $O
done
cat $O
