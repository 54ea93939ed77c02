"""The bench driver's command in isolation: one handle at cfg4 (or MxN), `warm` pivots, then calls of `steps` pivots each — a
budget that fits one block, i.e. the serial form: decisions, ONE sweep, fix-up.  Prints wall time per call; run it under
`rocprofv3 --kernel-trace` and look at the last call with scripts/trace_timeline.py.
    python scripts/one_block_call.py cfg4 20 5 4  [opt=value,...]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
opts = {}
for kv in (sys.argv[5] if len(sys.argv) > 5 else "").split(","):
    if kv.strip():
        k, v = kv.split("=")
        opts[k.strip()] = int(v)
rng = np.random.default_rng(1)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
st = lps.LPState(A, b, c, options=opts)
st.simplex_loop(max_pivots=warm)
for k in range(calls):
    t0 = time.perf_counter()
    status, piv, _ = st.simplex_loop(max_pivots=steps)
    dt = time.perf_counter() - t0
    print("call %d: %d pivots in %.3f ms = %.0f pivots/s (status %d)" % (k, piv, 1e3 * dt, piv / dt, status), flush=True)
inf = st.info()
print("block %d, sweep %s, chain %d wgs, overlapped %d" % (inf["block"], inf["sweep_kernel_name"], inf["chain_wgs"], inf["overlapped"]))
st.close()
