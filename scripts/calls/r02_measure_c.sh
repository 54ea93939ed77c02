#!/bin/bash
set -o pipefail
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 > gpurun_out/c1.json 2> gpurun_out/c1.err; python scripts/bench_line.py "alone_inplace default" < gpurun_out/c1.json | cut -c1-150
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 --option sweep_rows=128 > gpurun_out/c2.json 2> gpurun_out/c2.err; tail -3 gpurun_out/c2.err; python scripts/bench_line.py "alone_inplace rows128" < gpurun_out/c2.json | cut -c1-150
python bench.py --no-cpu-baseline --no-cfg3 --no-parity --steps 512 --option overlap=0 --option sweep_rows=32 > gpurun_out/c3.json 2> gpurun_out/c3.err; tail -3 gpurun_out/c3.err; python scripts/bench_line.py "alone_inplace rows32" < gpurun_out/c3.json | cut -c1-150
