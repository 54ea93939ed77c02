#!/bin/bash
# round 4, call m: the restructured fix-up kernel: parity (blocked, shards, multi), then the grids again
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -k "blocked or shard or multi_loop or ragged or 16_row or cfg5 or degenerate" > gpurun_out/r04_m_quick.log 2>&1
tail -4 gpurun_out/r04_m_quick.log
grep -q "failed\|error" gpurun_out/r04_m_quick.log && exit 1
timeout -k 10 400 python scripts/arith_grid.py cfg4 "fused=1;fused=1,block=64;fused=1,block=64,chain_cus=4;fused=0;fused=0,block=64" > gpurun_out/r04_m_grid_cfg4.txt 2>&1
cat gpurun_out/r04_m_grid_cfg4.txt
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=0;fused=1;fused=1,block=64" > gpurun_out/r04_m_grid_cfg3.txt 2>&1
cat gpurun_out/r04_m_grid_cfg3.txt
