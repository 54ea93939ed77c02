import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps
from oracle import pyoracle as oracle
m, n, block, ndev = 9, 2100, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(11 * m + n)
A, b, c = rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)
for rep in range(3):
    mt = lps.LPMulti(A, b, c, devices=[0] * ndev, block=block, options={"overlap": 1})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (1, block, 2 * block + 3, 0, 5 * block - 1, -1):
        t0 = time.time()
        try:
            st = mt.simplex_loop(max_pivots=budget)
        except Exception as ex:
            print("rep", rep, "budget", budget, "FAILED", ex, "%.2fs" % (time.time() - t0), mt.info()); break
        want = ref.simplex_loop(max_pivots=budget)
        gA, gb, gc, gv, gp = mt.read(); wA, wb, wc, wv, wp = ref.read()
        print("rep", rep, "budget", budget, st[:2], (want["status"], want["pivots"]), "bits", np.array_equal(gA.view(np.uint64), wA.view(np.uint64)) and gv == wv, mt.info()["chain_wgs"], flush=True)
    mt.close()
