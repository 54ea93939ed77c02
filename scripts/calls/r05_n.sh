#!/bin/bash
# round 5, call n: two grid questions with the round-5 decision kernel.  (1) the serial form (a budget of one block: the driver's
# command) at cfg4 with 128 workgroups — one row per thread on the whole chip — against 64; (2) "at most two columns per thread":
# 33 workgroups where the tableau is at most twice as wide as the rows the grid covers, repeated
mkdir -p gpurun_out
O=gpurun_out/r05_n.txt
: > $O
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_fine.so
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 overlap=0 chain_wgs=64 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 overlap=0 chain_wgs=128 2>&1 | tail -2 >> $O
timeout -k 10 120 python scripts/chain_trace_fine.py cfg4 256 overlap=0 chain_wgs=96 2>&1 | tail -2 >> $O
unset LPX_LIB_PATH
for rep in 1 2; do
timeout -k 10 400 python scripts/arith_grid.py cfg3 "block=0;chain_wgs=33;chain_wgs=41" 512 64 >> $O 2>&1
timeout -k 10 400 python scripts/arith_grid.py 4096x16384 "block=0;chain_wgs=33;chain_wgs=17" 512 64 >> $O 2>&1
timeout -k 10 400 python scripts/arith_grid.py 8192x8192 "block=0;chain_wgs=33;chain_wgs=17" 512 64 >> $O 2>&1
done
cat $O
