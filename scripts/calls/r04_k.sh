#!/bin/bash
# round 4, call k: k_sweep64_one (blocks of 33..64 by one wave per 64-column sub-strip): parity, then cfg4 / large grid
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged or blocks_of_64 or (blocked_pivoting_is_bit and (48 or 64 or 33)) or wide_decision" > gpurun_out/r04_k_quick.log 2>&1
tail -4 gpurun_out/r04_k_quick.log
grep -q "failed\|error" gpurun_out/r04_k_quick.log && exit 1
timeout -k 10 400 python scripts/arith_grid.py cfg4 "fused=1;fused=1,block=64;fused=1,block=64,sweep_form=2;fused=0;fused=0,block=64;fused=0,block=64,sweep_form=2;fused=1,block=64,chain_cus=12;fused=1,block=48" > gpurun_out/r04_k_grid_cfg4.txt 2>&1
cat gpurun_out/r04_k_grid_cfg4.txt
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=1;fused=1,block=64;fused=1,block=48" > gpurun_out/r04_k_grid_cfg3.txt 2>&1
cat gpurun_out/r04_k_grid_cfg3.txt
