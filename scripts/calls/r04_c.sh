#!/bin/bash
# round 4, call c: the new decision kernel (chain_form = 1): parity first (small tests, then the whole parity file in the
# default arithmetic), then same-box A/B of the two decision kernels
mkdir -p gpurun_out
export LPX_CHAIN_FORM=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "plain and (blocked or device_loop or wide_decision or by_size)" > gpurun_out/r04_c_quick.log 2>&1
tail -4 gpurun_out/r04_c_quick.log
grep -q "failed\|error" gpurun_out/r04_c_quick.log && exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "plain" > gpurun_out/r04_c_parity_form1.log 2>&1
tail -4 gpurun_out/r04_c_parity_form1.log
unset LPX_CHAIN_FORM
timeout -k 10 200 python scripts/arith_grid.py cfg3 "chain_form=0;chain_form=1;fused=1,chain_form=0;fused=1,chain_form=1;chain_form=0;chain_form=1" > gpurun_out/r04_c_grid_cfg3.txt 2>&1
cat gpurun_out/r04_c_grid_cfg3.txt
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=1,chain_form=0;fused=1,chain_form=1;fused=1,chain_form=1,chain_cus=8;fused=1,chain_form=1,chain_cus=8,block=64;fused=1,chain_form=0,chain_cus=8,block=64" > gpurun_out/r04_c_grid_cfg4.txt 2>&1
cat gpurun_out/r04_c_grid_cfg4.txt
for F in 0 1; do timeout -k 10 120 python scripts/chain_trace.py cfg3 256 chain_form=$F fused=1 2>&1 | tail -1; done > gpurun_out/r04_c_trace_cfg3.txt
cat gpurun_out/r04_c_trace_cfg3.txt
