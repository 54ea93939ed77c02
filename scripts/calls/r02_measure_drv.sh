#!/bin/bash
# the driver's bench command (20 timed pivots after 5: one partly filled block): zero-padded steady kernel against the
# tile kernel's guarded path for the partial block, same box, three repetitions each (the timed region is 2.4 ms)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked or forms or many_blocks or degenerate or resumed or cfg3_size" > gpurun_out/drv_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/drv_tests.log
T=$PWD/gpurun_variants/liblpx_tilespartial.so
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-110; }
for rep in 1 2 3; do
run "drv tiles " $T ""
run "drv padded" "" ""
done
