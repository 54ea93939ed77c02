#!/bin/bash
# round 4, call q: is the MFMA sweep bound by the traffic of its A operands (8 KiB of multipliers per 8 KiB tile, from
# L2 / Infinity Cache)?  LPX_SWEEP_DIAG=1: every tile reads tile 0's multipliers (results wrong, timing only)
mkdir -p gpurun_out
O=gpurun_out/r04_q.txt
: > $O
for D in 0 1; do
  export LPX_SWEEP_DIAG=$D
  echo "== LPX_SWEEP_DIAG=$D" >> $O
  timeout -k 10 200 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
done
cat $O
