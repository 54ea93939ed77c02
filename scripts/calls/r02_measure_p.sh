#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
run "new default" "" ""
run "prev default" $PWD/gpurun_variants/liblpx_prev.so ""
run "new alone" "" "--option overlap=0 --no-cfg3"
run "prev alone" $PWD/gpurun_variants/liblpx_prev.so "--option overlap=0 --no-cfg3"
for rows in 256 512 1024 2048 4096; do run "new alone rows=$rows" "" "--option overlap=0 --no-cfg3 --option sweep_rows=$rows"; done
run "new K16 alone" "" "--option overlap=0 --no-cfg3 --option block=16"
run "prev K16 alone" $PWD/gpurun_variants/liblpx_prev.so "--option overlap=0 --no-cfg3 --option block=16"
