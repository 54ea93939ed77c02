#!/bin/bash
# round 5, call j: all decision workgroups behind ONE L2 (experiment build -DLPX_CHAIN2_ONE_XCD: eight times the grid, only the
# workgroups dealt to XCD 0 take part) against the same grid spread over the eight XCDs; decisions alone (overlap = 0)
mkdir -p gpurun_out
O=gpurun_out/r05_j.txt
: > $O
for W in cfg3 4096x8192 cfg4; do
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
  echo "== spread over the XCDs, $W" >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 chain_wgs=32 2>&1 | tail -2 >> $O
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine_onexcd.so
  echo "== one XCD, $W" >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 overlap=0 chain_wgs=32 2>&1 | tail -2 >> $O
done
unset LPX_LIB_PATH
cat $O
