#!/bin/bash
# round 5, call v: the matrix-core sweep takes partly filled blocks itself (no generic launches behind it when the strips are
# whole) - parity of every blocked test in both arithmetics, then cfg4 / cfg3 throughput before (r05 library of the previous
# commit, kept as gpurun_variants/liblpx_prev.so) and after
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "block or 64 or cfg4 or by_size or ladders or beyond" > gpurun_out/r05_v_gpu.log 2>&1
tail -3 gpurun_out/r05_v_gpu.log
O=gpurun_out/r05_v_ab.txt
: > $O
for rep in 1 2 3; do
  for cfg in cfg4 cfg3; do
    echo "## prev $cfg" >> $O
    LPX_LIB_PATH=$PWD/gpurun_variants/liblpx_prev.so timeout -k 10 120 python scripts/arith_grid.py $cfg block=0 1024 64 >> $O 2>&1
    echo "## new $cfg" >> $O
    timeout -k 10 120 python scripts/arith_grid.py $cfg block=0 1024 64 >> $O 2>&1
  done
done
cat $O
