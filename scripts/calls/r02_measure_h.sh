#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/h_test.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/h_test.log
