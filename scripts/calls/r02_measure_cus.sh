#!/bin/bash
# CUs per XCD reserved for the decision kernel (4 = 32 CUs; 2 = 16; 3 = 24) with the steady-state sweep kernel, same box
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
for rep in 1 2; do
run "cfg4 4 CUs/XCD (33 wgs)" "" ""
run "cfg4 3 CUs/XCD (24 wgs)" $PWD/gpurun_variants/liblpx_cu3.so ""
run "cfg4 2 CUs/XCD (16 wgs)" $PWD/gpurun_variants/liblpx_cu2.so ""
done
