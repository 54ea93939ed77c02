#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked or timed_form or wide" > gpurun_out/e_test.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/e_test.log
python bench.py --no-cpu-baseline --no-parity > gpurun_out/e1.json 2> gpurun_out/e1.err; python scripts/bench_line.py "default" < gpurun_out/e1.json | cut -c1-130
python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 > gpurun_out/e2.json 2> gpurun_out/e2.err; python scripts/bench_line.py "driver" < gpurun_out/e2.json | cut -c1-130
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 > gpurun_out/e3.json 2> gpurun_out/e3.err; python scripts/bench_line.py "alone_inplace" < gpurun_out/e3.json | cut -c1-130
