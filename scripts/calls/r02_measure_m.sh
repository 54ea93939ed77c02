#!/bin/bash
# Final evidence of round 2: full GPU suite, bench lines, rocprof kernel stats and HBM traffic of the default bench, soak.
set -o pipefail
R=$PWD
OUT=$R/gpurun_out/prof_m
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > gpurun_out/m_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/m_gputest.log
python bench.py > gpurun_out/m_bench_default.json 2> gpurun_out/m_bench_default.err; echo "bench default rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/m_bench_driver.json 2> gpurun_out/m_bench_driver.err; echo "bench driver rc=$?"
python scripts/bench_line.py < gpurun_out/m_bench_default.json | cut -c1-400
python scripts/bench_line.py < gpurun_out/m_bench_driver.json | cut -c1-200
python scripts/soak_chain.py 90 > gpurun_out/m_soak.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/m_soak.txt
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 256 --warmup 64"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-cfg3 --no-parity > $OUT/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_traffic.py $OUT/fetch/*/*_counter_collection.csv $OUT/write/*/*_counter_collection.csv k_sweep32_steady 32768 16384 cfg4 32 > gpurun_out/m_traffic_cfg4_n1.json; cat gpurun_out/m_traffic_cfg4_n1.json | tail -8
cat $OUT/stats/*/*_kernel_stats.csv | cut -c1-160
