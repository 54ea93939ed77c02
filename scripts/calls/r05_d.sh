#!/bin/bash
# round 5, call d: 16-stamp traces only (a quick look at a ladder variant), then arith_grid at cfg3
mkdir -p gpurun_out
O=gpurun_out/r05_d.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
for W in cfg3 cfg4; do
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 fused=1 2>&1 | tail -2 >> $O
done
unset LPX_LIB_PATH
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;overlap=0" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=1" 512 64 >> $O 2>&1
cat $O
