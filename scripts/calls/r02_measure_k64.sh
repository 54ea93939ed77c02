#!/bin/bash
# blocks of 64 pivots (two-stage sweep + 64-slot decision kernel) against blocks of 32, same box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocked_pivoting_is_bit_identical or forms_are_bit_identical or many_blocks or degenerate_unbounded" > gpurun_out/k64_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/k64_tests.log
run() { python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 --workload $2 $3 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-120; }
for rep in 1 2; do
run "cfg4 block 32" cfg4 "--option block=32"
run "cfg4 block 64" cfg4 "--option block=64"
done
run "cfg3 block 32" cfg3 "--option block=32"
run "cfg3 block 64" cfg3 "--option block=64"
python scripts/chain_trace.py cfg4 256 block=64 > gpurun_out/k64_chain_cfg4.txt 2>&1; tail -3 gpurun_out/k64_chain_cfg4.txt
