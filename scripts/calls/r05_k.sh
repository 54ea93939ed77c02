#!/bin/bash
# round 5, call k: decision grid by size again with the round-5 kernel: 32 workgroups (one row per thread at cfg3, two columns
# per thread off the critical path) against 64, on 4 / 8 reserved CUs per XCD, both arithmetic modes
mkdir -p gpurun_out
O=gpurun_out/r05_k.txt
: > $O
timeout -k 10 400 python scripts/arith_grid.py cfg3 "fused=1;fused=1,chain_wgs=33;fused=1,chain_wgs=33,chain_cus=4;fused=1,chain_wgs=49;fused=0;fused=0,chain_wgs=33;fused=0,chain_wgs=33,chain_cus=4" 512 64 >> $O 2>&1
timeout -k 10 400 python scripts/arith_grid.py 6144x16384 "fused=1;fused=1,chain_wgs=33;fused=1,chain_wgs=33,chain_cus=4;fused=1,chain_wgs=25,chain_cus=4" 512 64 >> $O 2>&1
timeout -k 10 400 python scripts/arith_grid.py 12288x16384 "fused=1;fused=1,chain_wgs=49;fused=1,chain_wgs=33,chain_cus=4" 512 64 >> $O 2>&1
timeout -k 10 400 python scripts/arith_grid.py cfg4 "fused=1;fused=1,chain_wgs=33,chain_cus=4;fused=1,chain_wgs=33" 512 64 >> $O 2>&1
cat $O
