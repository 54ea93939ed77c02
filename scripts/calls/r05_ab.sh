#!/bin/bash
# round 5, call ab: one host round trip per blocked loop call (the loop state starts over inside the first launch and comes
# back in front of the loop's own final wait): whole GPU suite, then the driver's 20-pivot call in isolation and both bench lines
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_ab_gpu.log 2>&1
tail -3 gpurun_out/r05_ab_gpu.log
timeout -k 10 200 python scripts/one_block_call.py cfg4 20 5 8 > gpurun_out/r05_ab_one_block.txt 2>&1
cat gpurun_out/r05_ab_one_block.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_ab_bench_driver.json 2> gpurun_out/r05_ab_bench_driver.err
timeout -k 10 300 python bench.py > gpurun_out/r05_ab_bench_default.json 2> gpurun_out/r05_ab_bench_default.err
python - <<'PY'
import json
for f in ("gpurun_out/r05_ab_bench_driver.json", "gpurun_out/r05_ab_bench_default.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value %.0f ms_per_step %.5f sweep %.4f ms frac %.3f parity %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d.get("parity_after_timed_region", {}).get("ok")))
    for k in ("steady", "cfg3", "steady_plain"):
        if k in d: print("  ", k, json.dumps(d[k])[:300])
PY
