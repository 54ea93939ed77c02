#!/bin/bash
# round 4, call aj: what the selects inside the pending-pivot chains of k_block_chain2 cost (diagnostic build without them:
# wrong in a pending pivot's own row / column, timing only): 16 stamps per decision, with and without
mkdir -p gpurun_out
O=gpurun_out/r04_aj.txt
: > $O
for L in fine fine_nosel fine fine_nosel; do
  export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_$L.so
  echo "== $L" >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 overlap=0 2>&1 | tail -2 >> $O
  timeout -k 10 120 python scripts/chain_trace_fine.py cfg3 256 fused=1 chain_cus=8 2>&1 | tail -2 >> $O
done
cat $O
