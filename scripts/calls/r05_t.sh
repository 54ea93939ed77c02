#!/bin/bash
# round 5, call t: first-loop latency (ring creation + preparing launches) before / after the pruning of the product TU, then the
# whole GPU suite on the final build
mkdir -p gpurun_out
O=gpurun_out/r05_first_loop_latency.txt
: > $O
for rep in 1 2; do
  LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_r04.so timeout -k 10 120 python scripts/first_loop_latency.py >> $O 2>&1
  timeout -k 10 120 python scripts/first_loop_latency.py >> $O 2>&1
  LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_variants.so timeout -k 10 120 python scripts/first_loop_latency.py >> $O 2>&1
done
cat $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_t_gpu.log 2>&1
tail -5 gpurun_out/r05_t_gpu.log
