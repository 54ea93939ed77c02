#!/bin/bash
# round 4, call aq: the two bench commands on the final tree (bench.py: traffic of the fused legs, loop_bound labels)
R=$PWD
OUT=$R/gpurun_out/r04_aq
mkdir -p $OUT
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/drv.err | tail -1 > $OUT/bench_driver_command.json || { tail -20 $OUT/drv.err; }
python scripts/bench_line.py drv < $OUT/bench_driver_command.json
timeout -k 10 600 python bench.py 2>$OUT/def.err | tail -1 > $OUT/bench_default_cfg4.json || { tail -20 $OUT/def.err; }
python scripts/bench_line.py default < $OUT/bench_default_cfg4.json
