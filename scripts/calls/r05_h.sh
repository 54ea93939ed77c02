#!/bin/bash
# round 5, call h: fix-up with one chain per distinct slot / row: parity subset (both modes), multi rehearsal, same-box grid
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -k "degenerate or restart or blocked or ragged or wide_decision or cfg5 or cycling or decision or cfg3 or cfg4 or golden or spock or multi or shard" > gpurun_out/r05_h_quick.log 2>&1
tail -4 gpurun_out/r05_h_quick.log
O=gpurun_out/r05_h.txt
: > $O
timeout -k 10 300 python scripts/arith_grid.py cfg3 "fused=0;fused=1;block=0" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py cfg4 "fused=0;fused=1;block=0" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 6144x16384 "fused=0;fused=1;block=0" 512 64 >> $O 2>&1
timeout -k 10 300 python scripts/arith_grid.py 4096x16384 "fused=0;fused=1;block=0" 512 64 >> $O 2>&1
cat $O
