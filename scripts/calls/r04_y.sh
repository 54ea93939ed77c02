#!/bin/bash
# round 4, call y: k_sweep64_mfma2, third version (A operands by 16-byte loads, tickets taken a step late, one loop exit):
# the micro on 256 / 192 CUs, parity, grids
mkdir -p gpurun_out
O=gpurun_out/r04_y.txt
: > $O
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 32768 16384 10 32 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 32768 16384 10 24 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 8192 16384 20 24 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 4096 4096 20 24 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 1040 8192 20 32 >> $O 2>&1
cat $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "fused and (16_row_tiles or ragged or blocks_of_64 or wide_decision or cfg4 or beyond)" > gpurun_out/r04_y_quick.log 2>&1
tail -3 gpurun_out/r04_y_quick.log
timeout -k 10 200 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64;fused=1,block=64,chain_cus=4" 512 64 2>&1 | tee gpurun_out/r04_y_grid.txt
