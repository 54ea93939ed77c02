#!/bin/bash
# round 5, call z: the fix-up's chains beside the sweep in the serial (one-block) loop as well: parity, then the driver's command
# with the option off and on (same box)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixup or serial or degenerate or tracking or golden" > gpurun_out/r05_z_gpu.log 2>&1
tail -3 gpurun_out/r05_z_gpu.log
for rep in 1 2; do
  for mode in 0 2; do
    LPX_FIXUP_SIDE=$mode timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_z_driver_side$mode.json 2> gpurun_out/r05_z.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r05_z_driver_side$mode.json").read().strip().splitlines()[-1])
print("LPX_FIXUP_SIDE=$mode  value %.0f  ms_per_step %.5f  sweep %.4f ms  parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d.get("parity_after_timed_region", {}).get("ok")))
PY
  done
done
