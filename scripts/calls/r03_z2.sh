#!/bin/bash
# round 3, final code (by-size decision grid): the driver's command (twice) and the default command — full bench lines
set -o pipefail
OUT=gpurun_out/r03_z2; mkdir -p $OUT
s=$(date +%s); timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/drv.err | tail -1 > $OUT/bench_driver_command.json || { tail -5 $OUT/drv.err; exit 1; }
echo "driver command wall time: $(( $(date +%s) - s )) s"
python scripts/bench_line.py drv < $OUT/bench_driver_command.json
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_driver_command_2.json && python scripts/bench_line.py drv2 < $OUT/bench_driver_command_2.json
timeout -k 10 500 python bench.py 2>$OUT/def.err | tail -1 > $OUT/bench_default_cfg4.json || { tail -5 $OUT/def.err; exit 1; }
python scripts/bench_line.py default < $OUT/bench_default_cfg4.json
timeout -k 10 100 python scripts/chain_trace.py cfg3 256 2>&1 | tail -1
