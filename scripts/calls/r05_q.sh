#!/bin/bash
# round 5, call q: (1) the new tests of the round on the final build (by-size arithmetic, long ladders, variants library);
# (2) flake hunt of the multi-device path rehearsed on one GPU: 500 fresh four-shard handles (and 300 eight-shard ones)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -k "by_size or long_ladders or variants or superseded or pair_of_waves or first_mfma or round3" > gpurun_out/r05_q_tests.log 2>&1
tail -3 gpurun_out/r05_q_tests.log
O=gpurun_out/r05_flake_hunt_multi.txt
: > $O
GPU_MAX_HW_QUEUES=16 timeout -k 10 900 python scripts/flake_hunt_multi.py 2048 4096 4 500 40,70,33 >> $O 2>&1
GPU_MAX_HW_QUEUES=16 timeout -k 10 900 python scripts/flake_hunt_multi.py 4096 2048 8 300 64,31 >> $O 2>&1
GPU_MAX_HW_QUEUES=16 timeout -k 10 900 python scripts/flake_hunt_multi.py 8192 16384 4 40 70,40 >> $O 2>&1
tail -12 $O
