#!/bin/bash
# round 4, call am: same-box A/B of the release libraries: decision chain with selects (old) against per-step hit bits (new)
mkdir -p gpurun_out
O=gpurun_out/r04_am.txt
: > $O
for k in 1 2; do
  for L in old new; do
    if [ $L = old ]; then export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_oldchain.so; else unset LPX_LIB_PATH; fi
    echo "== $L" >> $O
    timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;overlap=0" 512 64 2>&1 | grep -v "^#" >> $O
    timeout -k 10 300 python scripts/arith_grid.py 2048x8192 "fused=0" 512 64 2>&1 | grep -v "^#" >> $O
  done
done
cat $O
