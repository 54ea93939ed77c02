#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or timed_form or wide or multi_loop or cfg5 or cfg2" > gpurun_out/x_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/x_test.log
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
for rep in 1 2; do
run "steady default" "" ""
run "nosteady default" $PWD/gpurun_variants/liblpx_nosteady.so ""
done
run "steady alone" "" "--option overlap=0 --no-cfg3"
run "nosteady alone" $PWD/gpurun_variants/liblpx_nosteady.so "--option overlap=0 --no-cfg3"
