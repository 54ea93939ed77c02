#!/bin/bash
# round 4, call t: what bounds k_sweep64_mfma2?  diagnostic builds (-DLPX_MFMA_DIAG=bits; results wrong, timing only):
# 2 no stores, 4 no tile loads, 8 no MFMAs (memory pass alone), 16 no A loads, 6 arithmetic + A alone, 20 MFMAs + stores
mkdir -p gpurun_out
O=gpurun_out/r04_t.txt
: > $O
for L in "" md2 md4 md8 md16 md6 md20; do
  if [ -n "$L" ]; then export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$L.so; fi
  echo "== lib ${L:-default}" >> $O
  timeout -k 10 150 python scripts/arith_grid.py cfg4 "fused=1,block=64,overlap=0;fused=1,block=64" 256 64 >> $O 2>&1
done
cat $O
