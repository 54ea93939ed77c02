#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 1024 --workload cfg3 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-80; }
for rep in 1 2; do
run "sweep 28 CUs/XCD (base)" "" ""
run "sweep 24" $PWD/gpurun_variants/liblpx_sw24.so ""
run "sweep 20" $PWD/gpurun_variants/liblpx_sw20.so ""
run "sweep 16" $PWD/gpurun_variants/liblpx_sw16.so ""
run "sweep 12" $PWD/gpurun_variants/liblpx_sw12.so ""
done
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_sw16.so python scripts/chain_trace.py cfg3 256 | tail -1
python scripts/chain_trace.py cfg3 256 | tail -1
