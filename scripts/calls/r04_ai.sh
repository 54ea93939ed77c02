#!/bin/bash
# round 4, call ai: soak of the blocked loop in the fused-arithmetic mode (random LPs in random budget pieces, every state
# bit-compared with the fused oracle): heights that are multiples of 16 (the MFMA sweep of blocks of 64) and ragged ones
mkdir -p gpurun_out
timeout -k 10 400 python scripts/soak_chain.py 150 1024x2112,2048x1536,512x4096,4096x640,300x700,513x515,1536x5000 -1 fused > gpurun_out/r04_soak_fused.txt 2>&1
tail -3 gpurun_out/r04_soak_fused.txt
