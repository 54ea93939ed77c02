#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
run "new(lb3) K16 alone" "" "--option overlap=0 --option block=16"
run "lb2 K16 alone" $PWD/gpurun_variants/liblpx_lb2.so "--option overlap=0 --option block=16"
run "prev K16 alone" $PWD/gpurun_variants/liblpx_prev.so "--option overlap=0 --option block=16"
run "new(lb3) K16 default" "" "--option block=16"
run "lb2 K16 default" $PWD/gpurun_variants/liblpx_lb2.so "--option block=16"
run "prev K16 default" $PWD/gpurun_variants/liblpx_prev.so "--option block=16"
for wl in cfg4_shard8 cfg4_shard2; do
LPX_LIB_PATH= python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 --workload $wl 2>/dev/null | python scripts/bench_line.py "new $wl" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_lb2.so python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 --workload $wl 2>/dev/null | python scripts/bench_line.py "lb2 $wl" | cut -c1-100
LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 --workload $wl 2>/dev/null | python scripts/bench_line.py "prev $wl" | cut -c1-100
done
