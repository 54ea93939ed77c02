#!/bin/bash
# round 3, first look: the LDS-DMA sweep kernel alone against the register one (three ring depths, cfg4 and cfg3
# sizes, nt on/off), then the blocked-loop parity tests through the library (default = the DMA kernel)
set -o pipefail
mkdir -p gpurun_out
for ns in 4 3 2; do
  echo "== ring slots $ns, cfg4, nt" ; timeout -k 10 120 scripts/micro/sweep_dma_ns$ns 32768 16384 10 0 1 || exit 1
done
echo "== ring slots 4, cfg4, default cache policy"; timeout -k 10 120 scripts/micro/sweep_dma_ns4 32768 16384 10 0 0 || exit 1
echo "== ring slots 4, cfg3, nt"; timeout -k 10 120 scripts/micro/sweep_dma_ns4 8192 16384 20 0 1 || exit 1
echo "== ring slots 4, cfg4 on 224 CUs' worth of rows, nt"; timeout -k 10 120 scripts/micro/sweep_dma_ns4 32768 16384 10 0 1 224 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blocked or timed_form or cfg3_size or wide_decision" 2>&1 | tail -15
