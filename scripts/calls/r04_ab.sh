#!/bin/bash
# round 4, call ab: k_sweep64_mfma2 with and without the nt hint on the tableau's loads and stores
mkdir -p gpurun_out
O=gpurun_out/r04_ab.txt
: > $O
for nt in 1 0 1 0; do
  echo "== nt $nt" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_bp0 32768 16384 10 24 $nt 2>&1 | grep "np 64  k_sweep64_mfma2\|flat" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_bp0 32768 16384 10 32 $nt 2>&1 | grep "np 64  k_sweep64_mfma2" >> $O
done
cat $O
