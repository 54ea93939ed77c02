#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-90; }
for rep in 1 2; do
run "base alone" "" "--option overlap=0"
run "D=1 alone" $PWD/gpurun_variants/liblpx_d1.so "--option overlap=0"
run "D=4 alone" $PWD/gpurun_variants/liblpx_d4.so "--option overlap=0"
run "D=6 alone" $PWD/gpurun_variants/liblpx_d6.so "--option overlap=0"
done
