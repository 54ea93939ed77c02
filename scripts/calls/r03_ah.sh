#!/bin/bash
# round 3: policies that predate the round-3 kernels, by size: overlapped vs serial blocks, nt vs cached streaming
S=1280x2048,2048x2048,2048x4096,4096x4096,4096x8192,8192x8192
echo "== default"; timeout -k 10 300 python scripts/block_policy.py $S 0 2>&1 | tail -6
echo "== overlap=0 (serial blocks)"; LPX_OVERLAP=0 timeout -k 10 300 python scripts/block_policy.py $S 0 2>&1 | tail -6
echo "== nt=0"; LPX_NT=0 timeout -k 10 300 python scripts/block_policy.py $S 0 2>&1 | tail -6
echo "== nt=1"; LPX_NT=1 timeout -k 10 300 python scripts/block_policy.py $S 0 2>&1 | tail -6
echo "== overlap=0, K=8/16/32"; LPX_OVERLAP=0 timeout -k 10 300 python scripts/block_policy.py $S 8,16,32 2>&1 | tail -6
