#!/bin/bash
# round 5, call ax (the round's last GPU seconds): k_sweep128_mfma (variants) against two passes of k_sweep64_mfma2, bit for bit; a
# small shape first, the cfg4 shape only if that one is right
mkdir -p gpurun_out
O=gpurun_out/r05_sweep_mfma128.txt
: > $O
timeout -k 5 30 scripts/micro/sweep_mfma128 1024 1024 2 32 2 >> $O 2>&1 && \
timeout -k 5 40 scripts/micro/sweep_mfma128 32768 16384 10 24 6 >> $O 2>&1
echo "rc=$?"
cat $O
