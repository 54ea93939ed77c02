#!/bin/bash
# round 4, call s: validation of the tree — GPU suite in both modes, smoke, the driver's command and the default command
R=$PWD
OUT=$R/gpurun_out/r04_s
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1
tail -3 $OUT/gputest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/drv.err | tail -1 > $OUT/bench_driver_command.json || { tail -20 $OUT/drv.err; }
python scripts/bench_line.py drv < $OUT/bench_driver_command.json
timeout -k 10 600 python bench.py 2>$OUT/def.err | tail -1 > $OUT/bench_default_cfg4.json || { tail -20 $OUT/def.err; }
python scripts/bench_line.py default < $OUT/bench_default_cfg4.json
