#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-105; }
run "base(4cu,33wg)" "" ""
run "cu2 wgs16" $PWD/gpurun_variants/liblpx_cu2.so "--option chain_wgs=16"
run "cu3 wgs24" $PWD/gpurun_variants/liblpx_cu3.so "--option chain_wgs=24"
run "cu1 wgs8" $PWD/gpurun_variants/liblpx_cu1.so "--option chain_wgs=8"
run "base rows128" "" "--option sweep_rows=128"
run "cu2 wgs16 rows128" $PWD/gpurun_variants/liblpx_cu2.so "--option chain_wgs=16 --option sweep_rows=128"
