#!/bin/bash
# full GPU suite with the opt-in blocks of 64 in the tree
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/k64_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/k64_suite.log
