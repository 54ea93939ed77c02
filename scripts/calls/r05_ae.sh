#!/bin/bash
# round 5, call ae: validation of the tree (whole GPU suite, smoke) and the round's evidence re-captured on it: kernel stats per
# workload, the two bench lines
R=$PWD
OUT=$R/gpurun_out/r05_ae
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -3 $OUT/gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg3 --no-steady --no-fused --no-onepass --no-parity > $OUT/stats_cfg4.log 2>&1; echo "stats cfg4 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg3 --no-cpu-baseline --no-onepass --no-parity > $OUT/stats_cfg3.log 2>&1; echo "stats cfg3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_steady -- python3 $R/bench.py --gpus 1 --steps 512 --warmup 64 --no-cpu-baseline --no-cfg3 --no-steady --no-onepass --no-parity > $OUT/stats_steady.log 2>&1; echo "stats steady rc=$?"
cd $R
for W in cfg4 cfg3 steady; do find $OUT/stats_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$W.csv; done
head -8 $OUT/kernel_stats_steady.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "driver bench rc=$?"
python scripts/bench_line.py < $OUT/bench_driver_command.json
timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
python scripts/bench_line.py < $OUT/bench_default.json
