#!/bin/bash
# round 4, call ao: soaks on the final code — default arithmetic (small and mid shapes), fused mid shapes with an LP cap
mkdir -p gpurun_out
timeout -k 10 300 python scripts/soak_chain.py 120 > gpurun_out/r04_soak_default.txt 2>&1
tail -2 gpurun_out/r04_soak_default.txt
timeout -k 10 400 python scripts/soak_chain.py 200 2048x4096,4096x4096,1024x8192 1500 fused > gpurun_out/r04_soak_fused_mid.txt 2>&1
tail -2 gpurun_out/r04_soak_fused_mid.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python scripts/soak_multi.py 120 > gpurun_out/r04_soak_multi.txt 2>&1
tail -2 gpurun_out/r04_soak_multi.txt
