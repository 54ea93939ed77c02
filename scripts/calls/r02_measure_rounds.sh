#!/bin/bash
# rows per run of k_sweep32_steady chosen for whole rounds of resident workgroups, against multiples of 48, same box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked or forms or many_blocks or timed_form or beyond" > gpurun_out/rounds_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/rounds_tests.log
P=$PWD/gpurun_variants/liblpx_rows48.so
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 1024 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-300; }
for rep in 1 2; do
run "rows x48     " $P ""
run "exact rounds " "" ""
done
run "drv rows x48    " $P "--steps 20 --warmup 5"
run "drv exact rounds" "" "--steps 20 --warmup 5"
