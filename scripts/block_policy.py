"""Pivots/s of lpx_simplex_loop for several tableau sizes and block sizes (checks the by-size policy of choose_block).
    python scripts/block_policy.py [MxN,MxN,...] [K,K,...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402

rng = np.random.default_rng(1)
shapes = ((1024, 2048), (2048, 2048), (2048, 4096), (4096, 4096), (4096, 8192), (8192, 8192))
if len(sys.argv) > 1:
    shapes = tuple(tuple(int(x) for x in sh.split("x")) for sh in sys.argv[1].split(","))
blocks = (0, 1, 8, 16, 32)
if len(sys.argv) > 2:
    blocks = tuple(int(x) for x in sys.argv[2].split(","))
for m, n in shapes:
    A = rng.random((m, n))
    b = (n / 4.0) * (1.0 + rng.random(m))
    c = rng.random(n)
    line = "%5d x %5d (%4d MiB):" % (m, n, m * n * 8 >> 20)
    for K in blocks:
        st = lps.LPState(A, b, c, block=K)
        st.simplex_loop(max_pivots=64)
        t0 = time.perf_counter()
        status, piv, _ = st.simplex_loop(max_pivots=1024)
        dt = time.perf_counter() - t0
        line += "  K=%s %7.0f" % ("auto(%d)" % st.block() if K == 0 else K, piv / dt)
        st.close()
    print(line, flush=True)
