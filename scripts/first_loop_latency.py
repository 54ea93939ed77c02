"""Latency of the first blocked loop of a process (ring creation + the one-workgroup preparing launch of every kernel the
blocked loop may select) and of the second handle's (ring creation alone):  python scripts/first_loop_latency.py [M N]
Compare libraries through LPX_LIB_PATH (e.g. gpurun_variants/liblpx_r04.so: round 4 launched ~60 kernels there)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 4096)
rng = np.random.default_rng(1)
A, b, c = rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)
out = []
for k in range(3):
    t0 = time.perf_counter()
    st = lps.LPState(A, b, c, block=32)
    t1 = time.perf_counter()
    st.simplex_loop(max_pivots=40)
    t2 = time.perf_counter()
    st.simplex_loop(max_pivots=40)
    t3 = time.perf_counter()
    st.close()
    out.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
print("%s  %dx%d: handle / first loop / second loop [ms]: %s" % (os.environ.get("LPX_LIB_PATH", "liblpx.so"), m, n,
                                                                 "  |  ".join("%.1f / %.1f / %.1f" % o for o in out)))
