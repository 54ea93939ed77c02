#!/bin/bash
# tagged candidate records in the decision kernel against the counter barrier + record read, same box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or wide or forms or loop or timed_form_200 or ties or degenerate" > gpurun_out/tag_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tag_tests.log
U=$PWD/gpurun_variants/liblpx_untagged.so
run() { LPX_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-parity --no-cfg3 --steps 1024 --workload $3 $4 2>/dev/null | python scripts/bench_line.py "$1" | cut -c1-100; }
for rep in 1 2; do
run "cfg3 counter" $U cfg3 ""
run "cfg3 tagged " "" cfg3 ""
done
run "cfg4 counter" $U cfg4 ""
run "cfg4 tagged " "" cfg4 ""
LPX_LIB_PATH=$U python scripts/chain_trace.py cfg3 256 | tail -1
python scripts/chain_trace.py cfg3 256 | tail -1
LPX_LIB_PATH=$U python scripts/chain_trace.py cfg3 256 overlap=0 | tail -1
python scripts/chain_trace.py cfg3 256 overlap=0 | tail -1
