#!/bin/bash
set -o pipefail
echo "== full kernel, cfg4"; timeout -k 10 120 scripts/micro/sweep_dma 32768 16384 10 0 1 | grep -v "ordered"
echo "== memory pass alone, cfg4"; timeout -k 10 120 scripts/micro/sweep_dma_diag1 32768 16384 10 0 1 | grep "np 32"| grep -v "ordered"
for cus in 240 224 208 192; do
echo "== full kernel, cfg4, $cus CUs"; timeout -k 10 120 scripts/micro/sweep_dma 32768 16384 10 0 1 $cus | grep "np 32"| grep -v "ordered"
done
echo "== memory pass alone, cfg4, 224 CUs"; timeout -k 10 120 scripts/micro/sweep_dma_diag1 32768 16384 10 0 1 224 | grep "np 32"| grep -v "ordered"
echo "== full kernel, cfg3"; timeout -k 10 120 scripts/micro/sweep_dma 8192 16384 20 0 1 | grep "np 32"| grep -v "ordered"
echo "== full kernel, 1000 x 2100"; timeout -k 10 120 scripts/micro/sweep_dma 1000 2100 20 0 1 | grep -v "ordered"
echo "== full kernel, 4100 x 1024"; timeout -k 10 120 scripts/micro/sweep_dma 4100 1024 20 0 1 | grep -v "ordered"
exit 0
