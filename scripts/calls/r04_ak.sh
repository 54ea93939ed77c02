#!/bin/bash
# round 4, call ak: k_block_chain2 with the pending pivots' parameters read unconditionally (no lgkmcnt(0) in front of every
# chunk): parity, 16-stamp traces (with and without the selects), same-box grid at cfg3 / cfg4 / mid sizes
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "degenerate or restart or blocked or ragged or wide_decision or cfg5 or cycling or decision" > gpurun_out/r04_ak_quick.log 2>&1
tail -3 gpurun_out/r04_ak_quick.log
O=gpurun_out/r04_ak.txt
: > $O
for L in fine fine_nosel; do
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$L.so
  echo "== $L" >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 overlap=0 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 fused=1 chain_cus=8 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 overlap=0 2>&1 | tail -2 >> $O
done
unset LPX_LIB_PATH
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;fused=1,block=64" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 4096x8192 "fused=0;fused=1" 512 64 >> $O 2>&1
cat $O
