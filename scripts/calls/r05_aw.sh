#!/bin/bash
# round 5, call aw (the round's last GPU-minutes): the arithmetic of a block of 128 pivots on the memory pass of a block of 64
# (scripts/micro/sweep_mfma_k128.hip: diagnostic copy of k_sweep64_mfma2 with its MFMA groups run twice per tile), 192 and 256 CUs
mkdir -p gpurun_out
O=gpurun_out/r05_sweep_mfma_k128.txt
: > $O
timeout -k 5 40 scripts/micro/sweep_mfma_k128_r1 32768 16384 10 24 >> $O 2>&1 && \
timeout -k 5 40 scripts/micro/sweep_mfma_k128_r2 32768 16384 10 24 >> $O 2>&1 && \
timeout -k 5 40 scripts/micro/sweep_mfma_k128_r2 32768 16384 10 32 >> $O 2>&1 && \
timeout -k 5 40 scripts/micro/sweep_mfma_k128_r2 8192 16384 10 24 >> $O 2>&1
echo "rc=$?"
cat $O
