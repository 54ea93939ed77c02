#!/bin/bash
# round 4, call ah: k_sweep64_mfma3 (16-row x 128-column tiles, two tile buffers) against k_sweep64_mfma2: micro, parity, grids
mkdir -p gpurun_out
O=gpurun_out/r04_ah.txt
: > $O
for keep in 24 32 24 32; do
  timeout -k 10 100 gpurun_variants/sweep_mfma_d0 32768 16384 10 $keep 2>&1 | grep -v "np 40" >> $O
done
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 8192 16384 20 24 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 4096 4096 20 24 >> $O 2>&1
timeout -k 10 100 gpurun_variants/sweep_mfma_d0 1040 8192 20 32 >> $O 2>&1
cat $O
timeout -k 10 200 python scripts/arith_grid.py cfg4 "fused=1,block=64;fused=1,block=64,sweep_form=5;fused=1,block=64;fused=1,block=64,sweep_form=5" 512 64 2>&1 | tee gpurun_out/r04_ah_grid.txt
