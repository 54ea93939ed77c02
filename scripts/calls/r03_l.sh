#!/bin/bash
# round 3: ragged-shape tests of the pull kernels; decision phase traces with the round-3 sweep beside them
set -o pipefail
mkdir -p gpurun_out/r03_l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged" 2>&1 | tail -5 || exit 1
timeout -k 10 200 python scripts/chain_trace.py cfg3 256 > gpurun_out/r03_l/chain_phases_cfg3.txt 2>&1; tail -2 gpurun_out/r03_l/chain_phases_cfg3.txt
timeout -k 10 200 python scripts/chain_trace.py cfg4 256 > gpurun_out/r03_l/chain_phases_cfg4.txt 2>&1; tail -2 gpurun_out/r03_l/chain_phases_cfg4.txt
timeout -k 10 200 python scripts/chain_trace.py cfg3 256 overlap=0 > gpurun_out/r03_l/chain_phases_cfg3_alone.txt 2>&1; tail -1 gpurun_out/r03_l/chain_phases_cfg3_alone.txt
