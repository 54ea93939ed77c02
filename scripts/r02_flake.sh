#!/bin/bash
# the full-size parity tests (cfg3 / cfg4 / > 4 GiB / blocks of 64) several times over in one process each
mkdir -p gpurun_out
for rep in 1 2 3 4 5 6; do
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "timed_form or beyond or blocks_of_64 or cfg3_size or cfg4_size or wide_decision" > gpurun_out/flake_big_$rep.log 2>&1
  echo "big tests rep $rep: $(tail -1 gpurun_out/flake_big_$rep.log)"
  grep -E "^FAILED|AssertionError: tableau" gpurun_out/flake_big_$rep.log | head -5
done
