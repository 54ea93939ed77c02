#!/bin/bash
# round 4, call e: k_block_chain2, second version (shorter compute path): parity, then traces and A/B
mkdir -p gpurun_out
export LPX_CHAIN_FORM=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "plain and (blocked or device_loop or wide_decision or by_size or dantzig or degenerate)" > gpurun_out/r04_e_quick.log 2>&1
tail -4 gpurun_out/r04_e_quick.log
grep -q "failed\|error" gpurun_out/r04_e_quick.log && exit 1
unset LPX_CHAIN_FORM
O=gpurun_out/r04_e_trace.txt
: > $O
for W in cfg3 cfg4; do
  for X in "overlap=0" "" "fused=1" "fused=1 chain_cus=8"; do
    timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -3 | grep -v "   mean" >> $O
  done
done
cat $O
timeout -k 10 200 python scripts/arith_grid.py cfg3 "chain_form=0;chain_form=1;fused=1,chain_form=0;fused=1,chain_form=1" > gpurun_out/r04_e_grid_cfg3.txt 2>&1
cat gpurun_out/r04_e_grid_cfg3.txt
export LPX_CHAIN_FORM=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q > gpurun_out/r04_e_full_form1.log 2>&1
tail -4 gpurun_out/r04_e_full_form1.log
