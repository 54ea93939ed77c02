import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps
from oracle import pyoracle as oracle
ndev = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 100)
rng = np.random.default_rng(11 * m + n)
A, b, c = rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)
mt = lps.LPMulti(A, b, c, devices=[0] * ndev, block=4 if m < 1000 else 32)
ref = oracle.State(A, b, c, kind=oracle.FP64)
t0 = time.time()
try:
    st = mt.simplex_loop(max_pivots=50)
    want = ref.simplex_loop(max_pivots=50)
    gA, gb, gc, gv, gp = mt.read(); wA, wb, wc, wv, wp = ref.read()
    print("ndev", ndev, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "->", st, want["status"], want["pivots"],
          "bits equal", np.array_equal(gA.view(np.uint64), wA.view(np.uint64)) and gv == wv and list(gp) == list(wp), "%.2fs" % (time.time() - t0), mt.info())
except Exception as ex:
    print("ndev", ndev, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "FAILED", ex, "%.2fs" % (time.time() - t0))
