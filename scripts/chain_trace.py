"""Phase durations of the decision kernel's last block (LPX_OPT_CHAIN_TRACE through the handle; 100 MHz ticks -> us)
plus the placement census: which XCDs the decision kernel and the sweep ran on, the grid and its residency bound.
    python scripts/chain_trace.py cfg3|cfg4|MxN [pivots=256] [name=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import linear_programming_solver_amd as lps  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    m, n = bench.WORKLOADS[wl] if wl in bench.WORKLOADS else map(int, wl.split("x"))
    pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    opts = {"chain_trace": 1}
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        opts[k] = int(v)
    A, b, c = bench.gen_rows(m, n, 1, 0, m)
    st = lps.LPState(A, b, c, options=opts)
    st.simplex_loop(max_pivots=64)
    status, piv, _ = st.simplex_loop(max_pivots=pivots - 1)   # -1: the last block is a full one plus nothing
    tr = st.chain_trace()
    info = st.info()
    print("# %s %dx%d options %s -> %s" % (wl, m, n, opts, info))
    print("s  phaseA  barrier1  phaseB  handoff/barrier2  total(us)")
    for k, (t0, t1, t2, t3, t4) in enumerate(tr):
        if not t0 or t4 < t3 or t3 < t2:   # a decision that ended the loop returns before the later stamps
            continue
        print("%2d %7.2f %8.2f %7.2f %8.2f %8.2f" % (k, (t1 - t0) / 100, (t2 - t1) / 100, (t3 - t2) / 100,
                                                       (t4 - t3) / 100, (t4 - t0) / 100))
    live = tr[(tr[:, 0] != 0) & (tr[:, 4] >= tr[:, 3]) & (tr[:, 3] >= tr[:, 2])]
    if len(live):
        d = (live[:, 4] - live[:, 0]) / 100.0
        print("block total %.1f us for %d decisions; per decision mean %.2f median %.2f us; phases mean A %.2f B1 %.2f "
              "B %.2f H %.2f" % ((live[-1, 4] - live[0, 0]) / 100, len(live), d.mean(), np.median(d),
                                 ((live[:, 1] - live[:, 0]) / 100).mean(), ((live[:, 2] - live[:, 1]) / 100).mean(),
                                 ((live[:, 3] - live[:, 2]) / 100).mean(), ((live[:, 4] - live[:, 3]) / 100).mean()))
    st.close()


if __name__ == "__main__":
    main()
