#!/bin/bash
set -o pipefail
export GPU_MAX_HW_QUEUES=16
for sh in 1 2 4 8; do
  python bench.py --no-cpu-baseline --steps 256 --rehearse-shards $sh > gpurun_out/i_rehearse_$sh.json 2> gpurun_out/i_rehearse_$sh.err; echo "rehearse $sh rc=$?"; tail -2 gpurun_out/i_rehearse_$sh.err | cut -c1-300; python scripts/bench_line.py "rehearse $sh" < gpurun_out/i_rehearse_$sh.json | cut -c1-330
done
python bench.py --no-cpu-baseline --steps 256 --rehearse-shards 2 --workload cfg3 2>/dev/null | python scripts/bench_line.py "cfg3 rehearse 2" | cut -c1-200
