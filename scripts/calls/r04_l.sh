#!/bin/bash
# round 4, call l: k_sweep64_mfma (fused blocks of 33..64 on the matrix cores): parity, then the cfg4 / cfg3 grids
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "16_row_tiles or (fused and (ragged or blocks_of_64 or wide_decision))" > gpurun_out/r04_l_quick.log 2>&1
tail -4 gpurun_out/r04_l_quick.log
grep -q "failed\|error" gpurun_out/r04_l_quick.log && exit 1
timeout -k 10 400 python scripts/arith_grid.py cfg4 "fused=1;fused=1,block=64;fused=1,block=64,sweep_form=3;fused=1,block=64,sweep_form=2;fused=1,block=64,chain_cus=4;fused=1,block=48" > gpurun_out/r04_l_grid_cfg4.txt 2>&1
cat gpurun_out/r04_l_grid_cfg4.txt
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=1;fused=1,block=64;fused=1,block=48" > gpurun_out/r04_l_grid_cfg3.txt 2>&1
cat gpurun_out/r04_l_grid_cfg3.txt
