#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or timed_form or wide or multi_loop" > gpurun_out/t_test.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t_test.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "new driver" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "prev driver" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_stamps.so python scripts/sweep_stamps.py > gpurun_out/t_wg_lifetimes.txt 2>&1; head -3 gpurun_out/t_wg_lifetimes.txt | cut -c1-300
