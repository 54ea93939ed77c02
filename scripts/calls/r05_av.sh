#!/bin/bash
# round 5, call av (the round's last GPU-minutes): do v_mfma_f64 and v_fma_f64 run side by side on a SIMD?  (scripts/micro/fp64_coexec.hip)
mkdir -p gpurun_out
timeout -k 5 60 scripts/micro/fp64_coexec > gpurun_out/r05_fp64_coexec.txt 2>&1; echo "rc=$?"
cat gpurun_out/r05_fp64_coexec.txt
