#!/bin/bash
# round 5, call e: the whole GPU suite (both arithmetic modes) on the windowed ladder without the window loop
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_e_gpu.log 2>&1
tail -5 gpurun_out/r05_e_gpu.log
