#!/bin/bash
# round 3: the driver's command (20 pivots after 5) by the grid of the decision kernel when nothing runs beside it:
# decision time alone at cfg4 by workgroups (serial form, whole chip), then the bench line with unmasked streams
for w in 33 64 96 128 192; do
  echo "== cfg4 alone chain_wgs=$w"; timeout -k 10 200 python scripts/chain_trace.py cfg4 96 overlap=0 chain_wgs=$w 2>&1 | tail -1
done
for o in "" "--option overlap_mask=0 --option chain_wgs=64" "--option overlap_mask=0 --option chain_wgs=128"; do
  for r in 1 2; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-steady --no-onepass --no-cfg3 --no-cpu-baseline --no-parity $o 2>/dev/null | python scripts/bench_line.py "drv[$o]"
  done
done
