#!/bin/bash
# round 4, call a: GPU tests in both arithmetic modes, then the first fused-vs-plain grid (same box)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gputest_b.log 2>&1
tail -5 gpurun_out/r04_gputest_b.log
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1;fused=1,block=64;fused=0,block=64;fused=1,chain_cus=8;fused=1,block=64,chain_cus=8;fused=1,chain_cus=16;fused=1,block=64,chain_cus=16;fused=0;fused=1" > gpurun_out/r04_grid_cfg4_a.txt 2>&1
cat gpurun_out/r04_grid_cfg4_a.txt
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=0;fused=1;fused=1,chain_cus=12;fused=1,chain_cus=16;fused=1,block=16;fused=1,block=64;fused=0;fused=1" > gpurun_out/r04_grid_cfg3_a.txt 2>&1
cat gpurun_out/r04_grid_cfg3_a.txt
