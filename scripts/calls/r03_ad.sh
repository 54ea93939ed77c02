#!/bin/bash
# round 3: the decision kernel's grid by size (one row / column per thread; 8 CUs per XCD for decision-bound tableaus
# above 8192 rows / columns) against the build before (gpurun_variants/liblpx_base.so), same box
B="--no-steady --no-onepass --no-cfg3 --no-cpu-baseline"
for v in base new base new; do
  if [ $v = base ]; then export LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_base.so; else unset LPX_LIB_PATH; fi
  echo "== [$v]"
  timeout -k 10 300 python scripts/block_policy.py 1024x2048,2048x2048,2048x4096,4096x4096,4096x8192,8192x8192,16384x8192,8192x16384,16384x16384 0 2>&1 | tail -9
  timeout -k 10 300 python bench.py --workload cfg3 $B 2>/dev/null | python scripts/bench_line.py "cfg3[$v]"
  timeout -k 10 300 python bench.py --workload cfg3 --steps 20 --warmup 5 $B 2>/dev/null | python scripts/bench_line.py "cfg3 drv[$v]"
  timeout -k 10 300 python bench.py $B --no-parity 2>/dev/null | python scripts/bench_line.py "cfg4[$v]"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 $B --no-parity 2>/dev/null | python scripts/bench_line.py "cfg4 drv[$v]"
done
