#!/bin/bash
# round 5, call m: the long-ladder test (second window of the 64-slot decision kernel) in both modes, then soaks on the final
# code: default loop (plain, small shapes), fused mid shapes with an LP cap, lpx_multi with 2-4 shards on one GPU
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "long_ladders or blocks_of_64" > gpurun_out/r05_m_tests.log 2>&1
tail -3 gpurun_out/r05_m_tests.log
timeout -k 10 300 python scripts/soak_chain.py 100 > gpurun_out/r05_soak_default.txt 2>&1
tail -2 gpurun_out/r05_soak_default.txt
timeout -k 10 400 python scripts/soak_chain.py 150 2048x4096,4096x4096,1024x8192 1500 fused > gpurun_out/r05_soak_fused_mid.txt 2>&1
tail -2 gpurun_out/r05_soak_fused_mid.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 400 python scripts/soak_multi.py 200 > gpurun_out/r05_soak_multi.txt 2>&1
tail -2 gpurun_out/r05_soak_multi.txt
