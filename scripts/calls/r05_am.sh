#!/bin/bash
# round 5, call am: the profiling events of a pulled sweep are the kernel's own start / stop events: whole GPU suite, the
# driver's one-block call, the two bench lines (do HIP-event kernel times still agree with rocprofv3's? kernel stats of the steady leg)
R=$PWD
OUT=$R/gpurun_out/r05_am
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -3 $OUT/gpu_suite.log
timeout -k 10 200 python scripts/one_block_call.py cfg4 20 5 8 > $OUT/one_block.txt 2>&1; tail -4 $OUT/one_block.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "driver bench rc=$?"
python scripts/bench_line.py < $OUT/bench_driver_command.json
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
python scripts/bench_line.py < $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_steady -- python3 $R/bench.py --gpus 1 --steps 512 --warmup 64 --no-cpu-baseline --no-cfg3 --no-steady --no-onepass --no-parity > $OUT/stats_steady.log 2>&1; echo "stats steady rc=$?"
cd $R
find $OUT/stats_steady -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_steady.csv
head -4 $OUT/kernel_stats_steady.csv | cut -c1-60,170-260
tail -c 1200 $OUT/stats_steady.log | grep -o '"avg_kernel_ms": [0-9.]*' | head -2
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
