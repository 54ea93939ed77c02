#!/bin/bash
# round 4, call o: k_sweep64_mfma2 (two waves per SIMD, B operands in LDS): parity, A/B against the one-wave form
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "16_row_tiles" > gpurun_out/r04_o_quick.log 2>&1
tail -4 gpurun_out/r04_o_quick.log
grep -q "failed\|error" gpurun_out/r04_o_quick.log && exit 1
timeout -k 10 400 python scripts/arith_grid.py cfg4 "fused=1,block=64;fused=1,block=64,sweep_form=4;fused=1,block=64,sweep_form=4,chain_cus=4;fused=1,block=64;fused=1,block=64,sweep_form=4" > gpurun_out/r04_o_grid_cfg4.txt 2>&1
cat gpurun_out/r04_o_grid_cfg4.txt
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=1,block=64;fused=1,block=64,sweep_form=4;fused=1" > gpurun_out/r04_o_grid_cfg3.txt 2>&1
cat gpurun_out/r04_o_grid_cfg3.txt
