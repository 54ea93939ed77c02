#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/w_test.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/w_test.log
export GPU_MAX_HW_QUEUES=32
for sh in 2 4 8; do
  python bench.py --no-cpu-baseline --steps 256 --rehearse-shards $sh > gpurun_out/w_rehearse_$sh.json 2> gpurun_out/w_rehearse_$sh.err; echo "rehearse $sh rc=$?"; tail -1 gpurun_out/w_rehearse_$sh.err | cut -c1-300; python scripts/bench_line.py "rehearse $sh (overlapped)" < gpurun_out/w_rehearse_$sh.json | cut -c1-250
done
python bench.py --no-cpu-baseline --steps 256 --rehearse-shards 8 --option overlap=0 2>/dev/null | python scripts/bench_line.py "rehearse 8 serial" | cut -c1-120
