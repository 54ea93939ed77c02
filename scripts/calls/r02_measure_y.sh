#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/y_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/y_test.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "driver" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_nosteady.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "driver nosteady" | cut -c1-100
python scripts/soak_chain.py 60 | tail -1
