#!/bin/bash
# round 5, call s: cfg4 through lpx_multi with 2 and 4 shards on ONE GPU (rehearsal of the multi-device path on the round-5
# decision kernel with the two-hop exchange): oracle replay in the line
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for N in 2 4; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 256 --rehearse-shards $N > gpurun_out/r05_rehearse_${N}shards_cfg4.json 2> gpurun_out/r05_rehearse_${N}shards_cfg4.err; echo "rehearse $N rc=$?"
  python scripts/bench_line.py < gpurun_out/r05_rehearse_${N}shards_cfg4.json | cut -c1-400
done
