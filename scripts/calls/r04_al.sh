#!/bin/bash
# round 4, call al: k_block_chain2 with per-step hit bits instead of selects: parity, 16-stamp traces, same-box grid
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -k "degenerate or restart or blocked or ragged or wide_decision or cfg5 or cycling or decision or cfg3 or golden or spock" > gpurun_out/r04_al_quick.log 2>&1
tail -3 gpurun_out/r04_al_quick.log
O=gpurun_out/r04_al.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 overlap=0 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 fused=1 chain_cus=8 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 overlap=0 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 fused=1 chain_cus=8 2>&1 | tail -2 >> $O
unset LPX_LIB_PATH
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;fused=1,block=64" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 4096x8192 "fused=0;fused=1" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 16384x16384 "fused=0;fused=1" 512 64 >> $O 2>&1
cat $O
