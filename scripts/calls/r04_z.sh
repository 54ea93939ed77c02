#!/bin/bash
# round 4, call z: does the power-of-two row pitch (ld = 16384 doubles = 128 KiB at cfg3 / cfg4) cost the pulled sweeps
# bandwidth?  the MFMA sweep and its copy alone (LPX_MFMA_DIAG=24) with the pitch skewed by 512 / 1024 / 1536 doubles
mkdir -p gpurun_out
O=gpurun_out/r04_z.txt
: > $O
for ld in 0 16896 17408 17920; do
  for keep in 32 24; do
    echo "== pitch $ld doubles" >> $O
    timeout -k 10 100 gpurun_variants/sweep_mfma_d0 32768 16384 10 $keep 1 $ld 2>&1 | grep -v "k_sweep64_mfma \|np 40" >> $O
    timeout -k 10 100 gpurun_variants/sweep_mfma_d24 32768 16384 10 $keep 1 $ld 2>&1 | grep "np 64  k_sweep64_mfma2" >> $O
  done
done
cat $O
