#!/bin/bash
set -o pipefail
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-110; }
for rep in 1 2; do
run "base alone" "" "--option overlap=0"
run "stagger alone" $PWD/gpurun_variants/liblpx_stagger.so "--option overlap=0"
run "prio alone" $PWD/gpurun_variants/liblpx_prio.so "--option overlap=0"
run "rb8 alone" $PWD/gpurun_variants/liblpx_rb8.so "--option overlap=0"
run "nb3 alone" $PWD/gpurun_variants/liblpx_nb3rb4.so "--option overlap=0"
done
run "base default" "" ""
run "stagger default" $PWD/gpurun_variants/liblpx_stagger.so ""
run "prio default" $PWD/gpurun_variants/liblpx_prio.so ""
run "rb8 default" $PWD/gpurun_variants/liblpx_rb8.so ""
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/g_power.txt 2>&1 &
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 4096 --option overlap=0 2>/dev/null | python scripts/bench_line.py "long alone" | cut -c1-110
wait
tail -25 gpurun_out/g_power.txt | cut -c1-300
