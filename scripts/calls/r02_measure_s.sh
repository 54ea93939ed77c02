#!/bin/bash
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s_test.log
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
run "new default" "" ""
run "prev default" $PWD/gpurun_variants/liblpx_prev.so ""
run "new K16 default" "" "--option block=16 --no-cfg3"
run "prev K16 default" $PWD/gpurun_variants/liblpx_prev.so "--option block=16 --no-cfg3"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "new driver" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python scripts/bench_line.py "prev driver" | cut -c1-100
python scripts/block_policy.py > gpurun_out/s_block_policy.txt 2>&1; cat gpurun_out/s_block_policy.txt
