"""The clock probe (lpx_state_info.sweep_clock_mhz) by placement of the pack kernel: python scripts/clock_probe_check.py cfg3|cfg4 [reps]
One handle per fixup_side in (0, 2); `reps` loops of 256 pivots each, the probe read after every loop."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402

SHAPES = {"cfg3": (8192, 16384), "cfg4": (32768, 16384)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m, n = SHAPES[name]
rng = np.random.default_rng(1)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
for side in (0, 2):
    st = lps.LPState(A, b, c, options={"fixup_side": side})
    st.simplex_loop(max_pivots=64)
    got = []
    for _ in range(reps):
        st.simplex_loop(max_pivots=256)
        got.append(st.info()["sweep_clock_mhz"])
    print("%s fixup_side=%d: sweep_clock_mhz %s" % (name, side, got), flush=True)
    st.close()
