#!/bin/bash
# Round-2 GPU measurement, part B: CU-mask census, sweep tile/np experiments, counter list.
set -o pipefail
R=$PWD
mkdir -p gpurun_out
scripts/micro/cu_mask > gpurun_out/r02b_cu_mask.txt 2>&1; echo "cu_mask rc=$?"; cat gpurun_out/r02b_cu_mask.txt
for rows in 32 64 128; do
  python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 --option sweep_rows=$rows 2>/dev/null | python scripts/bench_line.py "alone_inplace rows=$rows" | cut -c1-150
done
for st in 8 16 24 31; do
  python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps $st --warmup 33 --option overlap=0 2>/dev/null | python scripts/bench_line.py "alone np=$st" | cut -c1-150
done
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap_mask=0 2>/dev/null | python scripts/bench_line.py "overlap nomask" | cut -c1-150
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap_serial=1 2>/dev/null | python scripts/bench_line.py "overlap serial" | cut -c1-150
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option block=16 2>/dev/null | python scripts/bench_line.py "K16" | cut -c1-150
rocprofv3 -L > gpurun_out/r02b_counters.txt 2>&1; echo "rocprofv3 -L rc=$?"
