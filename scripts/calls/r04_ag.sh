#!/bin/bash
# round 4, call ag: k_sweep64_mfma2 with ONE tile in flight per wave (two tile buffers), two and three waves per SIMD
# (copy micro of round 3: the fewer tiles a worker keeps in flight, the better the pulled pattern streams)
mkdir -p gpurun_out
O=gpurun_out/r04_ag.txt
: > $O
for v in b3_t256 b2_t256 b2_t384 b3_t256 b2_t256 b2_t384; do
  echo "== $v" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_$v 32768 16384 10 24 2>&1 | grep "np 64  k_sweep64_mfma2\|flat" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_$v 32768 16384 10 32 2>&1 | grep "np 64  k_sweep64_mfma2" >> $O
  timeout -k 10 100 gpurun_variants/sweep_mfma_$v 8192 16384 20 24 2>&1 | grep "np 64  k_sweep64_mfma2" >> $O
done
cat $O
