#!/bin/bash
# round 4, call r: k_sweep64_mfma2 without the scheduling barriers between a tile's loads and the previous tile's MFMAs
mkdir -p gpurun_out
O=gpurun_out/r04_r.txt
: > $O
for L in "" gpurun_variants/liblpx_nosb.so; do
  if [ -n "$L" ]; then export LPX_LIB_PATH=$PWD/$L; fi
  echo "== lib ${L:-default}" >> $O
  timeout -k 10 200 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
done
cat $O
