#!/bin/bash
# round 3: CUs per XCD reserved for the decision kernel (option chain_cus: 4 = shipped, 6, 8), same box:
# cfg3 bench (twice), small shapes, cfg4
B="--no-steady --no-onepass --no-cfg3 --no-cpu-baseline"
for r in 1 2; do
  for c in 4 6 8; do
    timeout -k 10 300 python bench.py --workload cfg3 $B --option chain_cus=$c 2>/dev/null | python scripts/bench_line.py "cfg3[chain_cus=$c]"
  done
done
for c in 4 8; do
  timeout -k 10 300 python bench.py $B --no-parity --option chain_cus=$c 2>/dev/null | python scripts/bench_line.py "cfg4[chain_cus=$c]"
  LPX_CHAIN_CUS=$c timeout -k 10 200 python scripts/block_policy.py 2048x4096,4096x8192,8192x8192,16384x8192,8192x16384,16384x16384 0 2>&1 | tail -6
done
