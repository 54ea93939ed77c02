#!/bin/bash
# round 5, call as: where the fix-up / pack kernels should run in the DECISION-bound range (their default placement, the decisions'
# CUs, was chosen at cfg4 where the decisions have time to spare): fixup_side 2 / 1 / 0 at cfg3 (blocks of 64) and two mid sizes
mkdir -p gpurun_out
O=gpurun_out/r05_fixup_side_decision_bound.txt
: > $O
M="fixup_side=2;fixup_side=1;fixup_side=0;fixup_side=3;fixup_side=2;fixup_side=1;fixup_side=0;fixup_side=3"
for shape in cfg3 4096x8192 2048x4096 8192x8192; do
  timeout -k 10 200 python scripts/arith_grid.py $shape "$M" 2048 64 >> $O 2>&1
done
cat $O
