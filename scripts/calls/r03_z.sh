#!/bin/bash
# round 3, final code: GPU tests, smoke, the driver's command (twice) and the default command — full bench lines
set -o pipefail
OUT=gpurun_out/r03_z; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee $OUT/tests.txt || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt || exit 1
s=$(date +%s); timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/drv.err | tail -1 > $OUT/bench_driver_command.json || { tail -5 $OUT/drv.err; exit 1; }
echo "driver command wall time: $(( $(date +%s) - s )) s"
python scripts/bench_line.py drv < $OUT/bench_driver_command.json
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_driver_command_2.json && python scripts/bench_line.py drv2 < $OUT/bench_driver_command_2.json
timeout -k 10 500 python bench.py 2>$OUT/def.err | tail -1 > $OUT/bench_default_cfg4.json || { tail -5 $OUT/def.err; exit 1; }
python scripts/bench_line.py default < $OUT/bench_default_cfg4.json
