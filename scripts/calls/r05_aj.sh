#!/bin/bash
# round 5, call aj: the round's final validation: whole GPU suite, smoke, the two bench lines (now with roofline.bound = "mfma"
# and the roofline.mfma object for the matrix-core sweep), the two-shard rehearsal line
R=$PWD
OUT=$R/gpurun_out/r05_aj
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -3 $OUT/gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "driver bench rc=$?"
python scripts/bench_line.py < $OUT/bench_driver_command.json
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
python scripts/bench_line.py < $OUT/bench_default.json
GPU_MAX_HW_QUEUES=16 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 256 --rehearse-shards 2 > $OUT/rehearse_2shards.json 2> $OUT/rehearse_2shards.err; echo "rehearsal rc=$?"
python scripts/bench_line.py < $OUT/rehearse_2shards.json | head -5
