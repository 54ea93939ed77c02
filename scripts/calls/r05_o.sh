#!/bin/bash
# round 5, call o: the driver's command (one block of 20 pivots, fused by size at cfg4) with its sweep through the one-shot tile
# kernel k_update_tiles<32> (experiment build, sweep_form = 5) against the pulled kernel k_sweep32_pull
mkdir -p gpurun_out
O=gpurun_out/r05_o.txt
: > $O
for rep in 1 2; do
  echo "== pull (release)" >> $O
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg3 --no-steady --no-fused --no-onepass 2>/dev/null | python scripts/bench_line.py >> $O
  echo "== tiles (variant, sweep_form=5)" >> $O
  LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_tiles32.so timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cfg3 --no-steady --no-fused --no-onepass --option sweep_form=5 2>/dev/null | python scripts/bench_line.py >> $O
done
cat $O
