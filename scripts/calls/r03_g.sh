#!/bin/bash
# round 3: GPU tests after the ADVICE changes, the bench line with its new objects (driver command; rehearsal of the
# multi-GPU handle with 2 shards on this GPU), blocks of 64 against 32 at cfg4 with the round-3 sweep, hwmon paths
set -o pipefail
mkdir -p gpurun_out/r03_g
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40
cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap 2>&1 | head
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r03_g/drv.err | tail -1 > gpurun_out/r03_g/bench_drv.json || { tail -20 gpurun_out/r03_g/drv.err; exit 1; }
python scripts/bench_line.py drv < gpurun_out/r03_g/bench_drv.json
python -c "
import json; j=json.load(open('gpurun_out/r03_g/bench_drv.json')); print(json.dumps(j.get('onepass'))[:900]); print(j['roofline'].get('traffic'), j['roofline'].get('traffic_source')); print(j['steady']['cfg4'].get('board_power'))"
GPU_MAX_HW_QUEUES=8 timeout -k 10 400 python bench.py --rehearse-shards 2 --steps 96 --warmup 32 --no-cpu-baseline 2>gpurun_out/r03_g/reh.err | tail -1 > gpurun_out/r03_g/bench_rehearse2.json || { tail -20 gpurun_out/r03_g/reh.err; exit 1; }
python -c "
import json; j=json.load(open('gpurun_out/r03_g/bench_rehearse2.json')); print('rehearse 2 shards', j['value'], j['parity_after_timed_region'].get('ok'), json.dumps(j.get('onepass')))"
for blk in 32 64 32 64; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-cfg3 --no-onepass --option block=$blk 2>/dev/null | tail -1 | python scripts/bench_line.py block$blk
done
