#!/bin/bash
# round 4, call f: is the multi-GPU rehearsal still green after the warm-launch change (default decision kernel)?  three passes
mkdir -p gpurun_out
for k in 1 2 3; do
  timeout -k 10 400 python -m pytest tests/test_gpu_multi.py -q -k "plain" > gpurun_out/r04_f_multi_$k.log 2>&1
  tail -2 gpurun_out/r04_f_multi_$k.log
done
