#!/usr/bin/env python3
"""Sweeps the k_update tiling knobs (LPX_U, LPX_ROWS_PER_TILE, LPX_NT) on one GPU and prints the row-update
kernel's achieved GB/s for each (HIP-event timing through lpx_profile_*).  Usage: scripts/sweep_update.py cfg3"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
Us = [int(x) for x in os.environ.get("SWEEP_U", "1,2,4").split(",")]
Rs = [int(x) for x in os.environ.get("SWEEP_R", "4,8,16,32,64,128").split(",")]
NTs = [int(x) for x in os.environ.get("SWEEP_NT", "0,1").split(",")]
rows = []
for U, R, NT in itertools.product(Us, Rs, NTs):
    env = dict(os.environ, LPX_U=str(U), LPX_ROWS_PER_TILE=str(R), LPX_NT=str(NT))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "60",
                        "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        rows.append((U, R, NT, j["roofline"]["achieved"], j["roofline"]["avg_kernel_ms"], j["value"]))
        print("U=%d R=%3d NT=%d  k_update %.0f GB/s  %.4f ms   %.1f pivots/s" % rows[-1], flush=True)
    except Exception as ex:
        print("U=%d R=%d NT=%d failed: %s %s" % (U, R, NT, ex, p.stderr[-300:]), flush=True)
best = max(rows, key=lambda r: r[3])
print("best: U=%d R=%d NT=%d %.0f GB/s" % best[:4])
