#!/bin/bash
set -o pipefail
run() { LPX_EXPERIMENT_LD_PAD=$2 python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 512 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
for pad in 0 16 32 64 128 256 512; do run "alone pad=$pad" $pad "--option overlap=0"; done
for pad in 0 32 64; do run "default pad=$pad" $pad ""; done
for pad in 0 64; do LPX_EXPERIMENT_LD_PAD=$pad python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 200 --workload cfg3 --option block=1 2>/dev/null | python scripts/bench_line.py "cfg3 onepass pad=$pad" | cut -c1-100; done
