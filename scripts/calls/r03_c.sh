#!/bin/bash
# round 3: run form vs address-ordered form of the LDS-DMA sweep (alone; full kernel and its memory pass alone)
set -o pipefail
mkdir -p gpurun_out
for cus in 256 224; do
echo "== full kernel, cfg4, $cus CUs' worth of workgroups"; timeout -k 10 120 scripts/micro/sweep_dma 32768 16384 10 0 1 $cus
echo "== memory pass alone, cfg4, $cus"; timeout -k 10 120 scripts/micro/sweep_dma_diag1 32768 16384 10 0 1 $cus | grep "np 32"
done
echo "== full kernel, cfg3"; timeout -k 10 120 scripts/micro/sweep_dma 8192 16384 20 0 1 | grep "np 32"
echo "== full kernel, 4096 x 8192"; timeout -k 10 120 scripts/micro/sweep_dma 4096 8192 20 0 1
echo "== full kernel, 1000 x 2100 (partial strip, odd sizes)"; timeout -k 10 120 scripts/micro/sweep_dma 1000 2100 20 0 1
exit 0
