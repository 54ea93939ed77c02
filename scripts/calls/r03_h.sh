#!/bin/bash
# round 3: soaks with the pull kernel as the default sweep; CU share of the decision kernel at cfg3 (4 vs 8 per XCD)
set -o pipefail
mkdir -p gpurun_out/r03_h
LPX_SOAK=150 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k soak 2>&1 | tail -3
timeout -k 10 300 python scripts/soak_chain.py 90 2>&1 | tail -4
for v in default cu8 default cu8; do
  if [ $v = cu8 ]; then export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_cu8.so; OPT="--option chain_wgs=64"; else unset LPX_LIB_PATH; OPT=""; fi
  timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --no-onepass $OPT 2>/dev/null | tail -1 | python scripts/bench_line.py cfg3_$v
done
export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_cu8.so
timeout -k 10 300 python bench.py --workload cfg3 --no-cpu-baseline --no-onepass --option chain_wgs=48 2>/dev/null | tail -1 | python scripts/bench_line.py cfg3_cu8_48wgs
timeout -k 10 300 python bench.py --no-cpu-baseline --no-onepass --no-cfg3 --option chain_wgs=64 2>/dev/null | tail -1 | python scripts/bench_line.py cfg4_cu8_64wgs
