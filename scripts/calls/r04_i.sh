#!/bin/bash
# round 4, call i: k_block_chain2 with identity padding: parity, 16-stamp trace (diagnostic build), 8-stamp trace and A/B
mkdir -p gpurun_out
export LPX_CHAIN_FORM=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "plain and (blocked or device_loop or wide_decision or by_size or dantzig or degenerate or cfg5 or fuzz)" > gpurun_out/r04_i_quick.log 2>&1
tail -4 gpurun_out/r04_i_quick.log
grep -q "failed\|error" gpurun_out/r04_i_quick.log && exit 1
unset LPX_CHAIN_FORM
O=gpurun_out/r04_i_trace.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
for W in cfg3 cfg4; do
  for X in "overlap=0" "fused=1 chain_cus=8"; do
    timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -2 >> $O
  done
done
unset LPX_LIB_PATH
for W in cfg3 cfg4; do
  for X in "overlap=0" "" "fused=1" "fused=1 chain_cus=8"; do
    timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -3 | grep -v "   mean" >> $O
  done
done
cat $O
timeout -k 10 200 python scripts/arith_grid.py cfg3 "chain_form=0;chain_form=1;fused=1,chain_form=0;fused=1,chain_form=1" > gpurun_out/r04_i_grid_cfg3.txt 2>&1
cat gpurun_out/r04_i_grid_cfg3.txt
