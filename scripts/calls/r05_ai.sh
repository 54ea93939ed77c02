#!/bin/bash
# round 5, call ai: what the HIP events around every sweep launch (bench.py's live kernel timing) cost the loop they time:
# events around every launch / every 4th / none, cfg4 and cfg3, same box
mkdir -p gpurun_out
O=gpurun_out/r05_event_sampling_cost.txt
: > $O
for rep in 1 2; do
  for every in 1 4 0; do
    echo "## events around every ${every}th sweep launch (0 = none)" >> $O
    GRID_PROFILE_EVERY=$every timeout -k 10 200 python scripts/arith_grid.py cfg4 "block=0" 1024 64 2>&1 | grep pivots/s >> $O
    GRID_PROFILE_EVERY=$every timeout -k 10 200 python scripts/arith_grid.py cfg3 "block=0" 1024 64 2>&1 | grep pivots/s >> $O
  done
done
cat $O
