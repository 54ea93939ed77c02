#!/bin/bash
# round 3: the rehearsal line with its one-hop leg; wall time of the driver's command with everything in it
set -o pipefail
mkdir -p gpurun_out/r03_o
GPU_MAX_HW_QUEUES=8 timeout -k 10 400 python bench.py --rehearse-shards 2 --steps 96 --warmup 32 --no-cpu-baseline 2>gpurun_out/r03_o/reh.err | tail -1 > gpurun_out/r03_o/bench_rehearse2.json || { tail -20 gpurun_out/r03_o/reh.err; exit 1; }
python -c "
import json; j=json.load(open('gpurun_out/r03_o/bench_rehearse2.json')); print('rehearse 2 shards', round(j['value']), j['parity_after_timed_region'].get('ok'), j['parity_after_timed_region'].get('pivots_replayed'), json.dumps(j.get('onehop')), json.dumps(j.get('onepass'))[:200])"
T0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r03_o/drv.err | tail -1 > gpurun_out/r03_o/bench_drv.json
T1=$(date +%s.%N)
echo "driver command wall time: $(python -c "print(round($T1-$T0,1))") s"
python scripts/bench_line.py drv < gpurun_out/r03_o/bench_drv.json
python -c "
import json; j=json.load(open('gpurun_out/r03_o/bench_drv.json')); print(j['steady']['cfg4'].get('board_power')); print(j['roofline']['traffic'], j['devices_visible'])"
