#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
for rep in 1 2; do
run "base(rb4nb2) default" "" ""
run "rb2nb4 default" $PWD/gpurun_variants/liblpx_rb2nb4.so ""
run "rb2nb2 default" $PWD/gpurun_variants/liblpx_rb2nb2.so ""
run "rb4nb4 default" $PWD/gpurun_variants/liblpx_rb4nb4.so ""
done
run "base alone" "" "--option overlap=0"
run "rb2nb4 alone" $PWD/gpurun_variants/liblpx_rb2nb4.so "--option overlap=0"
