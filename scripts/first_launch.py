"""Why is the first 20-pivot call after a 5-pivot warm-up slower than later ones (the driver's bench command)?
Kernel instantiations used for the first time, or a GPU that has not yet clocked up?  Variant `busy`: the GPU is kept
busy with another kernel (the checksum) between the warm-up and the first timed call.
    python scripts/first_launch.py [cfg4|cfg3] [busy]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import linear_programming_solver_amd as lps  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
busy = len(sys.argv) > 2 and sys.argv[2] == "busy"
m, n = bench.WORKLOADS[wl]
A, b, c = bench.gen_rows(m, n, 1, 0, m)
st = lps.LPState(A, b, c)
st.simplex_loop(max_pivots=5)
if busy:
    for _ in range(20):
        st.checksum()
for rep in range(4):
    t0 = time.perf_counter()
    status, piv, _ = st.simplex_loop(max_pivots=20)
    dt = time.perf_counter() - t0
    print("%s%s call %d: 20 pivots in %.3f ms = %.0f pivots/s" % (wl, " busy" if busy else "", rep, dt * 1e3, piv / dt),
          flush=True)
