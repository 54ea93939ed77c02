#!/bin/bash
# round 5, call aq: the private ring copies of the decision kernel written THROUGH (st_agent) instead of left dirty in L2 until
# the launch ends (-DLPX_CHAIN2_OWN_WT): launch stamps (prologue, launch to launch) and pivots/s against the plain-store build
mkdir -p gpurun_out
O=gpurun_out/r05_own_copies_write_through.txt
: > $O
for lib in launchstamps ownwt; do
  echo "## library $lib" >> $O
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$lib.so
  timeout -k 10 100 python scripts/chain_launch_stamps.py cfg3 1024 2>&1 | tail -3 | cut -c1-230 >> $O
  timeout -k 10 100 python scripts/chain_launch_stamps.py 2048x4096 1024 2>&1 | tail -3 | cut -c1-230 >> $O
done
for rep in 1 2; do
  for lib in launchstamps ownwt; do
    echo "## library $lib" >> $O
    export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$lib.so
    timeout -k 10 100 python scripts/arith_grid.py cfg3 "block=0" 2048 64 2>&1 | grep pivots/s >> $O
    timeout -k 10 100 python scripts/arith_grid.py 2048x4096 "block=0" 2048 64 2>&1 | grep pivots/s >> $O
    timeout -k 10 100 python scripts/arith_grid.py 4096x8192 "block=0" 2048 64 2>&1 | grep pivots/s >> $O
  done
done
cat $O
