#!/bin/bash
# round 5, call af: soaks on the final tree (the fix-up's chains and the pack kernel beside the sweep on a third stream, one host
# round trip per call): default loop (plain, small shapes), fused mid shapes with an LP cap (blocks of 64 on the matrix cores),
# fused with blocks of 32 forced, lpx_multi with 2-4 shards on one GPU
mkdir -p gpurun_out
timeout -k 10 300 python scripts/soak_chain.py 110 > gpurun_out/r05_soak_final_default.txt 2>&1
tail -2 gpurun_out/r05_soak_final_default.txt
timeout -k 10 400 python scripts/soak_chain.py 160 2048x4096,4096x4096,1024x8192 1500 fused > gpurun_out/r05_soak_final_fused_mid.txt 2>&1
tail -2 gpurun_out/r05_soak_final_fused_mid.txt
LPX_BLOCK=32 timeout -k 10 300 python scripts/soak_chain.py 90 2048x4096,1024x8192,4096x2048 1500 fused > gpurun_out/r05_soak_final_fused_block32.txt 2>&1
tail -2 gpurun_out/r05_soak_final_fused_block32.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python scripts/soak_multi.py 100 > gpurun_out/r05_soak_final_multi.txt 2>&1
tail -2 gpurun_out/r05_soak_final_multi.txt
