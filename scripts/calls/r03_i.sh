#!/bin/bash
# round 3: blocks of 33..64 through k_sweep64_pull — parity tests that use blocks of 64, then same-box bench A/B
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "64 or blocked or wide_decision or beyond_4_gib" 2>&1 | tail -6 || exit 1
for cfg in "block=32" "block=64" "block=64 --option sweep_form=1" "block=32" "block=64"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-onepass --no-cfg3 --option $cfg 2>/dev/null | tail -1 | python scripts/bench_line.py "cfg4 $cfg"
done
