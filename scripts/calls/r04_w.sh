#!/bin/bash
# round 4, call w: the MFMA sweeps alone on a synthetic ring (scripts/micro/sweep_mfma), all CUs and 24 CUs per XCD (what
# the sweep has beside the decision kernel in fused mode); diagnostic builds: 2 no stores, 8 no MFMAs, 16 no A loads,
# 24 = the copy alone through this access pattern
mkdir -p gpurun_out
O=gpurun_out/r04_w.txt
: > $O
for d in 0 8 24 16 2; do
  for keep in 32 24; do
    timeout -k 10 100 gpurun_variants/sweep_mfma_d$d 32768 16384 10 $keep >> $O 2>&1
  done
done
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 8192 16384 20 32 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 8192 16384 20 24 >> $O 2>&1
cat $O
