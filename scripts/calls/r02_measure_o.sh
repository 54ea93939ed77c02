#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or timed_form or wide or multi or residency or cfg2 or fuzz" > gpurun_out/o_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/o_test.log
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
run "new default" "" ""
run "prev default" $PWD/gpurun_variants/liblpx_prev.so ""
run "new alone" "" "--option overlap=0 --no-cfg3"
run "prev alone" $PWD/gpurun_variants/liblpx_prev.so "--option overlap=0 --no-cfg3"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "new driver" | cut -c1-330
