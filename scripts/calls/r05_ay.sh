#!/bin/bash
# round 5, call ay: k_sweep128_mfma by row ranges per sub-strip (chunk size) on 192 CUs, and on 256 CUs
mkdir -p gpurun_out
O=gpurun_out/r05_sweep_mfma128_chunks.txt
: > $O
timeout -k 5 30 scripts/micro/sweep_mfma128 32768 16384 10 24 3 >> $O 2>&1 && \
timeout -k 5 30 scripts/micro/sweep_mfma128 32768 16384 10 24 12 >> $O 2>&1 && \
timeout -k 5 30 scripts/micro/sweep_mfma128 32768 16384 10 24 32 >> $O 2>&1 && \
timeout -k 5 30 scripts/micro/sweep_mfma128 32768 16384 10 32 8 >> $O 2>&1
echo "rc=$?"
grep -v "np 100\|np  64\|np  40" $O
