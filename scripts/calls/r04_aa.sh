#!/bin/bash
# round 4, call aa: k_sweep64_mfma2 with the B operands asked for one pivot group ahead (sched_group_barrier pipeline), A/B
mkdir -p gpurun_out
O=gpurun_out/r04_aa.txt
: > $O
for v in bp0 bp1 bp0 bp1; do
  echo "== $v" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_$v 32768 16384 10 24 2>&1 | grep "np 64  k_sweep64_mfma2\|flat" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_$v 32768 16384 10 32 2>&1 | grep "np 64  k_sweep64_mfma2" >> $O
done
cat $O
