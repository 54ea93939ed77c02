"""Soak test of lpx_multi with every "device" = device 0 (row-block shards exchanging candidates and pivot rows through
the peer mailboxes of k_block_chain2_t<..., MG = true>): random LPs solved in random budget pieces by 2-4 shards, both host loops
(decisions beside the sweeps / serial), every intermediate state compared bit for bit with the fp64 oracle.
    GPU_MAX_HW_QUEUES=16 python scripts/soak_multi.py [seconds=120]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # several shards share one GPU here: one hardware queue per stream
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def main():
    budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t_end = time.time() + budget_s
    rng = np.random.default_rng(20261004)
    lps.set_default_arithmetic("plain")   # (explicit: checked against the fp64 oracle)
    shapes = [(300, 700), (700, 1100), (1100, 260), (64, 2100), (513, 515)]
    n_lp = n_cmp = pivots_total = 0
    while time.time() < t_end:
        m, n = shapes[n_lp % len(shapes)]
        block = int(rng.choice([8, 16, 32]))
        shards = int(rng.choice([2, 3, 4]))
        A = rng.random((m, n))
        b = (n / 4.0) * (1.0 + rng.random(m))
        c = rng.random(n)
        opts = {"overlap": int(n_lp % 2)}
        mt = lps.LPMulti(A, b, c, devices=[0] * shards, block=block, options=opts)
        ref = oracle.State(A, b, c, kind=oracle.FP64)
        while True:
            budget = int(rng.choice([-1, 1, block, 3 * block + 1, 257, 1000]))
            got = mt.simplex_loop(max_pivots=budget)
            want = ref.simplex_loop(max_pivots=budget)
            assert (got[0], got[1]) == (want["status"], want["pivots"]), (n_lp, m, n, block, shards, budget, got, want)
            gA, gb, gc, gv, gp = mt.read()
            wA, wb, wc, wv, wp = ref.read()
            ok = (np.array_equal(bits(gA), bits(wA)) and np.array_equal(bits(gb), bits(wb)) and
                  np.array_equal(bits(gc), bits(wc)) and bits(np.array([gv]))[0] == bits(np.array([wv]))[0] and
                  np.array_equal(gp, wp))
            n_cmp += 1
            pivots_total += got[1]
            if not ok:
                print("MISMATCH lp %d shape %dx%d block %d shards %d budget %d" % (n_lp, m, n, block, shards, budget))
                sys.exit(1)
            if got[0] != 9:
                break
        mt.close()
        n_lp += 1
    print("multi soak ok: %d LPs, %d state comparisons, %d pivots, all bit-identical" % (n_lp, n_cmp, pivots_total))


if __name__ == "__main__":
    main()
