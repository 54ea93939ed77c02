#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "blocked or timed_form or wide or multi or cfg5 or fuzz or resumed" > gpurun_out/v_test.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/v_test.log
for rep in 1 2; do
for v in "" prev; do
  lib=""; [ -n "$v" ] && lib=$PWD/gpurun_variants/liblpx_$v.so
  echo "== variant '${v:-new}' rep $rep"
  LPX_LIB_PATH=$lib python scripts/chain_trace.py cfg3 256 2>&1 | tail -1
  LPX_LIB_PATH=$lib python scripts/chain_trace.py cfg3 256 overlap=0 2>&1 | tail -1
  LPX_LIB_PATH=$lib python scripts/chain_trace.py cfg4 256 2>&1 | tail -1
  LPX_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-parity --steps 512 --workload cfg3 2>/dev/null | python scripts/bench_line.py "cfg3" | cut -c1-60
done
done
