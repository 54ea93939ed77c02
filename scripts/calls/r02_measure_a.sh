#!/bin/bash
# Round-2 GPU measurement, part A: GPU tests, the default bench line, the driver's bench command, dp_rate, chain traces.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_gputest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_gputest.log
tail -5 gpurun_out/r02a_gputest.log
python bench.py > gpurun_out/r02a_bench_default.json 2> gpurun_out/r02a_bench_default.err; echo "bench default rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a_bench_driver.json 2> gpurun_out/r02a_bench_driver.err; echo "bench driver rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/dp_rate.hip -o scripts/micro/dp_rate && scripts/micro/dp_rate > gpurun_out/r02a_dp_rate.txt 2>&1; echo "dp_rate rc=$?"
python scripts/chain_trace.py cfg3 256 > gpurun_out/r02a_chain_cfg3.txt 2>&1; echo "trace cfg3 rc=$?"
python scripts/chain_trace.py cfg4 256 > gpurun_out/r02a_chain_cfg4.txt 2>&1; echo "trace cfg4 rc=$?"
python scripts/chain_trace.py cfg3 256 overlap=0 > gpurun_out/r02a_chain_cfg3_alone.txt 2>&1; echo "trace cfg3 alone rc=$?"
python scripts/bench_line.py < gpurun_out/r02a_bench_default.json
python scripts/bench_line.py < gpurun_out/r02a_bench_driver.json
