#!/bin/bash
# round 5, call au (the round's last GPU-minutes): the driver's own command on HEAD
mkdir -p gpurun_out/r05_au
timeout -k 10 260 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_au/bench_driver.json 2> gpurun_out/r05_au/bench_driver.err; echo "bench rc=$?"
python scripts/bench_line.py < gpurun_out/r05_au/bench_driver.json | head -14
