#!/bin/bash
# round 5, call at (re-entry, 9 GPU-minutes left): HEAD after the by-size fix-up placement (LPX_OPT_FIXUP_SIDE = 4) and the
# Makefile fix: the tests that run the loops whose placement changed (everything below 2 GiB now keeps the fix-up behind the
# sweep), the reference vectors, smoke, then the default bench line without its CPU legs if time is left
R=$PWD
OUT=$R/gpurun_out/r05_at
mkdir -p $OUT
K="fixup_beside or cfg3_timed or cfg4_timed or blocked_loop_forms or reference_ or cfg2_full_solve_matches or blocks_of_64_with_long or block_and_decision_grid or arithmetic_by_size or io_files or blocked_pivoting_is_bit"
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $OUT/gpu_subset.log 2>&1; echo "gpu subset rc=$?"
tail -3 $OUT/gpu_subset.log
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout -k 10 150 python bench.py --no-cpu-baseline --no-onepass > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python scripts/bench_line.py < $OUT/bench_default.json | head -12
