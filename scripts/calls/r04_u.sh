#!/bin/bash
# round 4, call u: k_sweep64_mfma2 with buffer addressing, hand-issued tickets and a straight-line loop: parity, then grids
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "fused and (16_row_tiles or ragged or blocks_of_64 or wide_decision or cfg4 or beyond)" > gpurun_out/r04_u_quick.log 2>&1
tail -3 gpurun_out/r04_u_quick.log
O=gpurun_out/r04_u_grid.txt
: > $O
timeout -k 10 200 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64;fused=1,block=64,chain_cus=4;fused=1,block=64,sweep_form=4" 512 64 >> $O 2>&1
timeout -k 10 200 python scripts/arith_grid.py cfg3 "fused=1;fused=1,block=64;fused=1,block=64,overlap=0" 512 64 >> $O 2>&1
cat $O
