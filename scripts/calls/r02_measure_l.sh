#!/bin/bash
set -o pipefail
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
