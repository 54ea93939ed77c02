#!/bin/bash
# round 5, call i: workgroups of 512 threads for the decision kernel (one row per thread at cfg4): parity at cfg4, traces, grid
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg4 or wide_decision or blocked" > gpurun_out/r05_i_quick.log 2>&1
tail -4 gpurun_out/r05_i_quick.log
O=gpurun_out/r05_i.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 fused=1 chain_threads=256 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 fused=1 chain_threads=512 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 fused=1 chain_threads=512 chain_cus=4 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 fused=0 chain_threads=512 2>&1 | tail -2 >> $O
unset LPX_LIB_PATH
timeout -k 10 300 python scripts/arith_grid.py cfg4 "chain_threads=256;chain_threads=512;chain_threads=512,chain_cus=4;fused=0,chain_threads=256;fused=0,chain_threads=512;fused=0,chain_threads=512,chain_cus=8" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 16384x16384 "chain_threads=256;chain_threads=512;chain_threads=512,chain_cus=4" 512 64 >> $O 2>&1
cat $O
