#!/bin/bash
# round 4, call h: 16 stamps per decision (diagnostic build): where inside the phases the time goes
mkdir -p gpurun_out
O=gpurun_out/r04_h_trace16.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
for W in cfg3 cfg4; do
  for X in "overlap=0" "fused=1 chain_cus=8"; do
    timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -2 >> $O
  done
done
cat $O
