#!/bin/bash
# the whole GPU suite several times over: any intermittent failure left?
mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/flake_suite_$rep.log 2>&1
  echo "suite rep $rep: $(tail -1 gpurun_out/flake_suite_$rep.log)"
  grep -E "^FAILED" gpurun_out/flake_suite_$rep.log | head -5
done
