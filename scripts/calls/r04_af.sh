#!/bin/bash
# round 4, call af: k_sweep32_pull with 24 steps per entry for blocks of at most 24 pivots: parity (budgets that end in
# short blocks), then the driver's command twice
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged or cfg4 or cfg3 or budget or blocked or sweep_forms or tail" > gpurun_out/r04_af_quick.log 2>&1
tail -3 gpurun_out/r04_af_quick.log
for k in 1 2; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-steady --no-fused --no-onepass 2>gpurun_out/r04_af.err | tail -1 > gpurun_out/r04_af_drv$k.json
  python scripts/bench_line.py drv$k < gpurun_out/r04_af_drv$k.json | cut -c1-400
done
