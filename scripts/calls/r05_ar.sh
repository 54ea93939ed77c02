#!/bin/bash
# round 5, call ar: last validation of the round's tree: whole GPU suite, smoke, soaks under the new block policy (mid shapes now
# take blocks of 32; fused mid shapes; 2-4 shards), the two-shard rehearsal line
R=$PWD
OUT=$R/gpurun_out/r05_ar
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"
tail -3 $OUT/gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 10 200 python scripts/soak_chain.py 70 2048x2048,2048x4096,1024x8192,4096x4096 1200 > $OUT/soak_mid_default.txt 2>&1; tail -1 $OUT/soak_mid_default.txt
timeout -k 10 200 python scripts/soak_chain.py 70 2048x4096,4096x4096,1024x8192 1200 fused > $OUT/soak_mid_fused.txt 2>&1; tail -1 $OUT/soak_mid_fused.txt
timeout -k 10 200 python scripts/soak_chain.py 50 > $OUT/soak_small_default.txt 2>&1; tail -1 $OUT/soak_small_default.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python scripts/soak_multi.py 60 > $OUT/soak_multi.txt 2>&1; tail -1 $OUT/soak_multi.txt
GPU_MAX_HW_QUEUES=16 timeout -k 10 600 python bench.py --no-cpu-baseline --steps 256 --rehearse-shards 2 > $OUT/rehearse_2shards.json 2> $OUT/rehearse_2shards.err; echo "rehearsal rc=$?"
python scripts/bench_line.py < $OUT/rehearse_2shards.json | head -3
