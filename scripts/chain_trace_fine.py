"""The eight stamps per decision of k_block_chain2 (option chain_form = 1) of the last block, as mean segment durations:
    python scripts/chain_trace_fine.py cfg3|cfg4|MxN [pivots=256] [name=value ...]
segments: A.loads (start -> phase A's loads here), A.chain (-> candidate published), X (-> every candidate read),
B.loads (-> phase B's loads here), B.chain (-> hand-off stored), B.rest (-> phase B done), H (-> next slot known)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import linear_programming_solver_amd as lps  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
m, n = bench.WORKLOADS[wl] if wl in bench.WORKLOADS else map(int, wl.split("x"))
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
opts = {"chain_trace": 1, "chain_form": 1}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    opts[k] = int(v)
A, b, c = bench.gen_rows(m, n, 1, 0, m)
st = lps.LPState(A, b, c, options=opts)
st.simplex_loop(max_pivots=64)
st.simplex_loop(max_pivots=pivots - 1)
tr = st.chain_trace_fine()
info = st.info()
names = ["A.loads", "A.chain", "X", "B.loads", "B.chain", "B.rest", "H"]
last = 7 if tr.shape[1] == 16 else tr.shape[1] - 1     # (16 stamps: slots 0..7 are the coarse ones, 7 the decision's last)
live = tr[(tr[:, 0] != 0) & (tr[:, last] >= tr[:, last - 1])]
seg = np.diff(live, axis=1) / 100.0
tot = (live[:, last] - live[:, 0]) / 100.0
print("%s %dx%d %s: grid %d (XCD mask 0x%02x), %d decisions, per decision mean %.2f median %.2f us" % (
    wl, m, n, {k: v for k, v in opts.items() if k != "chain_trace"}, info["chain_wgs"], info["chain_xcd_mask"], len(live), tot.mean(),
    np.median(tot)))
if tr.shape[1] == 16:   # diagnostic build (-DLPX_CHAIN2_FINE): stamps in time order
    order = [0, 1, 8, 9, 10, 11, 2, 3, 12, 4, 13, 14, 15, 5, 6, 7]
    nm16 = ["A.loads", "A.meet", "A.chain", "A.store+ratio", "A.min", "A.win", "X", "B.min+checks", "B.loads", "B.meet",
            "B.chain", "B.div+stores", "B.min", "B.rest", "H"]
    lv = live[:, order]
    sg = np.diff(lv, axis=1) / 100.0
    print("   median " + "  ".join("%s %.2f" % (nm, x) for nm, x in zip(nm16, np.median(sg, axis=0))))
elif seg.shape[1] == 7:
    print("   mean   " + "  ".join("%s %.2f" % (nm, x) for nm, x in zip(names, seg.mean(axis=0))))
    print("   median " + "  ".join("%s %.2f" % (nm, x) for nm, x in zip(names, np.median(seg, axis=0))))
else:
    print("   (5 stamps) mean segments", seg.mean(axis=0))
st.close()
