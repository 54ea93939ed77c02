#!/bin/bash
# round 4, call b: GPU tests in both arithmetic modes; the driver's bench command with the new legs
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gputest_b.log 2>&1
tail -5 gpurun_out/r04_gputest_b.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r04_bench_drv.err | tail -1 > gpurun_out/r04_bench_driver_command.json
python scripts/bench_line.py drv < gpurun_out/r04_bench_driver_command.json
tail -3 gpurun_out/r04_bench_drv.err
