#!/bin/bash
# round 5, call r: both candidate granules of a thread polled at once (64 workgroups): parity subset, traces, same-box A/B
# against the library without it (gpurun_variants/liblpx_head.so)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "degenerate or restart or blocked or wide_decision or cycling or decision or cfg3 or spock" > gpurun_out/r05_r_quick.log 2>&1
tail -3 gpurun_out/r05_r_quick.log
O=gpurun_out/r05_r.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 overlap=0 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 fused=1 2>&1 | tail -2 >> $O
unset LPX_LIB_PATH
for rep in 1 2; do
for L in head new; do
  echo "== $L" >> $O
  if [ $L = head ]; then export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_head.so; else unset LPX_LIB_PATH; fi
  timeout -k 10 300 python scripts/arith_grid.py cfg3 "block=0;fused=0;overlap=0" 512 64 >> $O 2>&1
  timeout -k 10 300 python scripts/arith_grid.py 12288x16384 "block=0" 512 64 >> $O 2>&1
done
done
unset LPX_LIB_PATH
cat $O
