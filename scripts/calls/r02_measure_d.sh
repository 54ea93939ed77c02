#!/bin/bash
set -o pipefail
python bench.py --no-cpu-baseline --no-parity > gpurun_out/d1.json 2> gpurun_out/d1.err; python scripts/bench_line.py "default" < gpurun_out/d1.json | cut -c1-400
python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 > gpurun_out/d2.json 2> gpurun_out/d2.err; python scripts/bench_line.py "driver" < gpurun_out/d2.json | cut -c1-200
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 > gpurun_out/d3.json 2> gpurun_out/d3.err; python scripts/bench_line.py "alone_inplace" < gpurun_out/d3.json | cut -c1-150
python scripts/chain_trace.py cfg3 256 > gpurun_out/d_chain_cfg3.txt 2>&1; tail -2 gpurun_out/d_chain_cfg3.txt
python scripts/chain_trace.py cfg4 256 > gpurun_out/d_chain_cfg4.txt 2>&1; tail -2 gpurun_out/d_chain_cfg4.txt
