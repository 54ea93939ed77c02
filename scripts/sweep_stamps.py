"""Diagnostic (needs a -DLPX_SWEEP_STAMPS=1 build of liblpx, LPX_LIB_PATH): start / end times of sampled sweep
workgroups — how the grid's rounds fill the chip."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import linear_programming_solver_amd as lps  # noqa: E402
from linear_programming_solver_amd import _lib  # noqa: E402

m, n = bench.WORKLOADS["cfg4"]
A, b, c = bench.gen_rows(m, n, 1, 0, m)
opts = {"overlap": 0}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    opts[k] = int(v)
st = lps.LPState(A, b, c, options=opts)
st.simplex_loop(max_pivots=64)
st.simplex_loop(max_pivots=96)
L = _lib.lib()
L.lpx_debug_read_census.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int32]
buf = np.zeros(256 + 2 + 1200, dtype=np.uint32)
assert L.lpx_debug_read_census(st._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size) == 0
raw = buf[256 + 8:256 + 8 + 140 * 4].view(np.int64).reshape(140, 2)
ok = raw[:, 0] != 0
t = raw[ok] / 100.0
t0 = t[:, 0].min()
print("options", opts, "info", st.info())
print("sampled workgroups %d: start min %.1f max %.1f us; end min %.1f max %.1f us; lifetime mean %.1f min %.1f max %.1f us" % (
    ok.sum(), 0.0, t[:, 0].max() - t0, t[:, 1].min() - t0, t[:, 1].max() - t0,
    (t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).min(), (t[:, 1] - t[:, 0]).max()))
order = np.argsort(t[:, 0])
print("start -> end (us) of the sampled workgroups in start order:")
print(" ".join("%.0f-%.0f" % (t[i, 0] - t0, t[i, 1] - t0) for i in order))
st.close()
