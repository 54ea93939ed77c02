#!/bin/bash
# round 4, call j: k_block_chain2 as the default decision kernel of the one-device loop: the whole GPU suite, both modes
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r04_j_gputest.log 2>&1
tail -6 gpurun_out/r04_j_gputest.log
