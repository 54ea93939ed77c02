#!/bin/bash
# round 3: the grid of the decision kernel at 4 CUs per XCD (one row / column per thread instead of two) and 6 / 8 CUs
timeout -k 10 120 python bench.py --workload cfg3 --no-steady --no-onepass --no-cfg3 --no-cpu-baseline --no-parity --option chain_cus=6 2>&1 | tail -3 | cut -c1-300
for w in 0 17 25 33; do
  echo "== chain_cus=4 chain_wgs=$w"
  LPX_CHAIN_WGS=$w timeout -k 10 200 python scripts/block_policy.py 2048x4096,4096x4096,4096x8192,8192x8192 0 2>&1 | tail -4
done
echo "== chain_cus=8"
LPX_CHAIN_CUS=8 timeout -k 10 200 python scripts/block_policy.py 2048x4096,4096x4096,4096x8192,8192x8192 0 2>&1 | tail -4
