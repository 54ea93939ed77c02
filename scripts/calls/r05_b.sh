#!/bin/bash
# round 5, call b: k_block_chain2 with the branch-free ladder (chain8 / chain8_from): parity subset in both modes, 16-stamp traces
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "degenerate or restart or blocked or ragged or wide_decision or cfg5 or cycling or decision or cfg3 or golden or spock" > gpurun_out/r05_b_quick.log 2>&1
tail -5 gpurun_out/r05_b_quick.log
O=gpurun_out/r05_b.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
for W in cfg3 cfg4; do
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 fused=1 2>&1 | tail -2 >> $O
done
unset LPX_LIB_PATH
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;overlap=0" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1" 512 64 >> $O 2>&1
cat $O
