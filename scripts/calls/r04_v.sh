#!/bin/bash
# round 4, call v: the second k_sweep64_mfma2 (buffer addressing, no drains) — what bounds it now?  diagnostic builds
# (results wrong, timing only): 2 no stores, 8 no MFMAs, 16 no A loads, 24 neither (the copy alone); LPX_SWEEP_DIAG=1:
# every tile reads tile 0's A operands
mkdir -p gpurun_out
O=gpurun_out/r04_v.txt
: > $O
echo "== default" >> $O
timeout -k 10 150 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
echo "== default, LPX_SWEEP_DIAG=1" >> $O
LPX_SWEEP_DIAG=1 timeout -k 10 150 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
for L in md2 md8 md16 md24; do
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$L.so
  echo "== lib $L" >> $O
  timeout -k 10 150 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
done
cat $O
