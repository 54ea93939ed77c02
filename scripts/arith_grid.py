"""Steady-state pivots/s of the default loop by arithmetic mode, block size and decision-kernel share, same box:
    python scripts/arith_grid.py cfg4|cfg3|MxN  "fused=0,block=32,chain_cus=4;fused=1,block=64,chain_cus=8;..."  [steps] [warmup]
Each `;`-separated item is a set of handle options (names: linear_programming_solver_amd._lib.OPTIONS).  One fresh handle
per item on the same tableau; prints pivots/s, ms per block, the mean sweep launch (HIP events) and what the engine did."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402

SHAPES = {"cfg2": (1024, 2048), "cfg3": (8192, 16384), "cfg4": (32768, 16384)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
m, n = SHAPES[name] if name in SHAPES else tuple(int(x) for x in name.split("x"))
items = (sys.argv[2] if len(sys.argv) > 2 else "fused=0;fused=1").split(";")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 64

rng = np.random.default_rng(1)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
print("# %s %dx%d, %d pivots after %d warm-up" % (name, m, n, steps, warm), flush=True)
for item in items:
    opts = {}
    for kv in item.split(","):
        if kv.strip():
            k, v = kv.split("=")
            opts[k.strip()] = int(v)
    st = lps.LPState(A, b, c, options=opts)
    st.simplex_loop(max_pivots=warm)
    st.profile_enable(int(os.environ.get("GRID_PROFILE_EVERY", "1")))   # HIP events around every n-th sweep launch (0: none)
    t0 = time.perf_counter()
    status, piv, _ = st.simplex_loop(max_pivots=steps)
    dt = time.perf_counter() - t0
    ln, ms = st.profile_read()
    st.profile_enable(False)
    inf = st.info()
    K = inf["block"]
    print("%-44s %8.0f pivots/s  block %2d = %.3f ms  sweep %.3f ms x%d (%s)  chain %d wgs%s" % (
        item, piv / dt, K, 1e3 * dt / piv * K, ms / ln if ln else float("nan"), ln, inf["sweep_kernel_name"],
        inf["chain_wgs"], " masked" if inf["chain_stream_masked"] else ""), flush=True)
    st.close()
