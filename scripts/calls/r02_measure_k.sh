#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or timed_form or wide or multi or residency" > gpurun_out/k_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/k_test.log
python bench.py --no-cpu-baseline --no-parity --steps 512 2>/dev/null | python scripts/bench_line.py "default" | cut -c1-110
python scripts/chain_trace.py cfg3 256 > gpurun_out/k_chain_cfg3.txt 2>&1; tail -1 gpurun_out/k_chain_cfg3.txt
python scripts/chain_trace.py cfg4 256 > gpurun_out/k_chain_cfg4.txt 2>&1; tail -1 gpurun_out/k_chain_cfg4.txt
python scripts/chain_trace.py cfg3 256 overlap=0 > gpurun_out/k_chain_cfg3_alone.txt 2>&1; tail -1 gpurun_out/k_chain_cfg3_alone.txt
