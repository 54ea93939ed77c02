#!/bin/bash
# round 5, call l: the whole GPU suite after the pruning of the product TU (variants library built beside it)
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r05_l_gpu.log 2>&1
tail -6 gpurun_out/r05_l_gpu.log
