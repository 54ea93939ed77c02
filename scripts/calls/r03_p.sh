#!/bin/bash
# round 3: decision time alone (serial form, whole chip) by the number of workgroups of the decision kernel, cfg3
for w in 8 16 24 32 33 48 64 96; do
  echo "== chain_wgs=$w"; timeout -k 10 100 python scripts/chain_trace.py cfg3 256 overlap=0 chain_wgs=$w 2>&1 | tail -1
done
