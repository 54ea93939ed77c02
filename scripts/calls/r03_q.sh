#!/bin/bash
# round 3: does a sweep sized for fewer CUs (less memory pressure) let the decisions beside it run faster?  cfg3
for c in 0 192 160 128 96 64; do
  timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-onepass --no-parity --option sweep_cus=$c 2>/dev/null | tail -1 | python scripts/bench_line.py "cfg3 sweep_cus=$c" | cut -c1-120
done
for c in 0 192 160; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-onepass --no-parity --no-cfg3 --option sweep_cus=$c 2>/dev/null | tail -1 | python scripts/bench_line.py "cfg4 sweep_cus=$c" | cut -c1-120
done
