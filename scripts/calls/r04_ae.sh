#!/bin/bash
# round 4, call ae: cfg3 is decision-bound and a decision takes 1.7 x longer beside the sweep than alone: does a sweep on
# fewer CUs (less pressure on the memory system) shorten the decisions by more than it lengthens the sweep?
mkdir -p gpurun_out
O=gpurun_out/r04_ae.txt
: > $O
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=1;fused=1,sweep_cus=160;fused=1,sweep_cus=128;fused=1,sweep_cus=96;fused=0;fused=0,sweep_cus=160;fused=0,sweep_cus=128" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 16384x16384 "fused=1;fused=1,sweep_cus=160;fused=0;fused=0,sweep_cus=160" 512 64 >> $O 2>&1
cat $O
