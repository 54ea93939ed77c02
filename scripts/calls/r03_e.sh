#!/bin/bash
# round 3: GPU tests with the pull kernel as the default sweep, then same-box A/B of the three sweep forms in bench.py
set -o pipefail
mkdir -p gpurun_out/r03_e
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 || exit 1
for form in 0 1 2 0; do
  echo "== bench default, sweep_form=$form"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-cfg3 --option sweep_form=$form 2>/dev/null | tail -1 > gpurun_out/r03_e/bench_form$form.json
  python scripts/bench_line.py < gpurun_out/r03_e/bench_form$form.json
done
for form in 0 1; do
  echo "== bench driver command, sweep_form=$form"
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --option sweep_form=$form 2>/dev/null | tail -1 > gpurun_out/r03_e/bench_drv_form$form.json
  python scripts/bench_line.py < gpurun_out/r03_e/bench_drv_form$form.json
done
for form in 0 1; do
  echo "== bench cfg3 1024 steps, sweep_form=$form"
  timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --option sweep_form=$form 2>/dev/null | tail -1 > gpurun_out/r03_e/bench_cfg3_form$form.json
  python scripts/bench_line.py < gpurun_out/r03_e/bench_cfg3_form$form.json
done
