#!/bin/bash
# the driver's bench command (20 timed pivots after 5: one partly filled block): overlapped form (nothing to overlap
# with) against the serial in-place form on the whole chip, same box
set -o pipefail
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 $2 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-110; }
for rep in 1 2; do
run "drv default  " ""
run "drv overlap=0" "--option overlap=0"
run "drv overlap=0 wgs=128" "--option overlap=0 --option chain_wgs=128"
done
