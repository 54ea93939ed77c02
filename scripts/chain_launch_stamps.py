"""Where a block of decisions spends its time OUTSIDE the decisions (diagnostic build -DLPX_CHAIN2_LAUNCH_STAMPS,
scripts/build_variant.sh launchstamps "-DLPX_CHAIN2_LAUNCH_STAMPS"; LPX_LIB_PATH=gpurun_variants/liblpx_launchstamps.so):
the decision kernel stamps its entry, the start of its first decision and of its last one; with the per-decision stamps
(option chain_trace) that gives, for the LAST two launches of a loop: prologue (entry -> first decision), the decisions,
and launch to launch (end of the last decision of launch k - 1 -> entry of launch k is not visible: the older launch's
per-decision stamps are overwritten; its LAST-decision START is kept, so the figure below is from there).
    python scripts/chain_launch_stamps.py cfg3|MxN [pivots] [opt=value,...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402
from linear_programming_solver_amd import _lib  # noqa: E402

SHAPES = {"cfg3": (8192, 16384), "cfg4": (32768, 16384)}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
m, n = SHAPES[name] if name in SHAPES else tuple(int(x) for x in name.split("x"))
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 512
opts = {"chain_trace": 1}
for kv in (sys.argv[3] if len(sys.argv) > 3 else "").split(","):
    if kv.strip():
        k, v = kv.split("=")
        opts[k.strip()] = int(v)
rng = np.random.default_rng(1)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
st = lps.LPState(A, b, c, options=opts)
L = _lib.lib()
L.lpx_debug_read_chain_dbg.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
st.simplex_loop(max_pivots=64)
K = st.info()["block"]
rows = []
for rep in range(6):
    # a budget that ends with a full block and no probing launch behind it: K * q - 1 pivots => the last launch decides K
    st.simplex_loop(max_pivots=(pivots // K) * K - 1)
    buf = np.zeros(1024, dtype=np.int64)
    assert L.lpx_debug_read_chain_dbg(st._h, buf.ctypes.data_as(C.POINTER(C.c_int64)), buf.size) == 0
    sets = [buf[512 + 8 * p: 512 + 8 * p + 4] for p in (0, 1)]
    new, old = (sets[0], sets[1]) if sets[0][3] > sets[1][3] else (sets[1], sets[0])
    nd = st.info()["block"]
    dec = buf[:8 * nd].reshape(nd, 8)
    per_dec = np.median(np.diff(dec[:, 0])) / 100.0
    rows.append(((new[1] - new[0]) / 100.0, per_dec, (dec[-1, 7] - dec[0, 0]) / 100.0, (new[0] - old[2]) / 100.0, int(new[3] - old[3])))
st.close()
print("%s %dx%d block %d %s" % (name, m, n, K, opts))
for r in rows:
    print("  prologue %6.2f us   decision (median) %6.2f us   all decisions of the launch %8.2f us   start of the previous launch's"
          " last decision -> entry of this launch %6.2f us (launch %+d)" % r)
