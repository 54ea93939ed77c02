#!/bin/bash
# round 5, call g: the two-hop exchange grafted onto k_block_chain2 (shards of an lpx_multi): multi-GPU rehearsal tests on one
# GPU in both arithmetic modes, then the kernel-trace timeline of the default cfg4 loop (r05_f)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/r05_g_multi.log 2>&1
tail -5 gpurun_out/r05_g_multi.log
bash scripts/calls/r05_f.sh
