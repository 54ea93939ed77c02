#!/bin/bash
# round 4, call g: what would a contiguous read of the entering column buy (diagnostic: LPX_CHAIN_DIAG=1, results wrong)?
mkdir -p gpurun_out
O=gpurun_out/r04_g_trace.txt
: > $O
for D in 0 1; do
  export LPX_CHAIN_DIAG=$D
  echo "== LPX_CHAIN_DIAG=$D" >> $O
  for W in cfg3 cfg4; do
    for X in "overlap=0" "fused=1 chain_cus=8"; do
      timeout -k 10 120 python scripts/chain_trace_fine.py $W 256 $X 2>&1 | tail -3 | grep -v "   mean" >> $O
    done
  done
done
cat $O
unset LPX_CHAIN_DIAG
export LPX_CHAIN_FORM=1
for k in 1 2 3; do
  timeout -k 10 400 python -m pytest tests/test_gpu_multi.py -q -k "plain" > gpurun_out/r04_g_multi_form1_$k.log 2>&1
  tail -2 gpurun_out/r04_g_multi_form1_$k.log
done
