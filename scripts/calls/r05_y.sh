#!/bin/bash
# round 5, call y: whole GPU suite on the build with the fix-up's chains beside the sweep (default LPX_OPT_FIXUP_SIDE = 2), then
# both bench lines
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_y_gpu.log 2>&1
tail -3 gpurun_out/r05_y_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/r05_y_bench_default.json 2> gpurun_out/r05_y_bench_default.err && tail -c 1500 gpurun_out/r05_y_bench_default.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_y_bench_driver.json 2> gpurun_out/r05_y_bench_driver.err && tail -c 600 gpurun_out/r05_y_bench_driver.json
