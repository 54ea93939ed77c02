#!/bin/bash
# round 4, call x: the copy alone through k_sweep64_mfma2's ticket structure (LPX_MFMA_DIAG=24) by shape of one memory
# instruction: 0 = 4 rows x 128 B (dwordx2, the MFMA's C layout), 1 = 4 rows x 256 B (dwordx4), 2 = 2 rows x 512 B
# (dwordx4), 3 = 1 row x 512 B (dwordx2)
mkdir -p gpurun_out
O=gpurun_out/r04_x.txt
: > $O
for sh in 0 1 2 3; do
  for keep in 32 24; do
    echo "== shape $sh" >> $O
    timeout -k 10 100 gpurun_variants/sweep_mfma_d24s$sh 32768 16384 10 $keep 2>&1 | grep -v "k_sweep64_mfma \|np 40" >> $O
  done
done
cat $O
