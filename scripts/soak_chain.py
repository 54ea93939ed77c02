"""Soak test of the blocked loop's default form (persistent decision kernel beside out-of-place sweeps): many random
LPs solved to optimality in random budget pieces, every intermediate state compared bit for bit with the fp64
oracle.  A stale read between workgroups of k_block_chain would show up here as a mismatch.
    python scripts/soak_chain.py [seconds=120] [MxN,MxN,...] [stop an LP after this many pivots] [fused]
"fused": handles in the fused-arithmetic mode against the oracle's fused instantiation (heights that are multiples of 16
send blocks of 33..64 through the matrix cores, k_sweep64_mfma2)."""
import os
import sys
import time

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
    rng = np.random.default_rng(20261003)
    shapes = [(300, 700), (700, 1100), (1100, 260), (64, 2100), (513, 515), (900, 900)]
    if len(sys.argv) > 2:   # "MxN,MxN,...": e.g. mid-size shapes, where the by-size grids of the decision kernel apply
        shapes = [tuple(int(x) for x in sh.split("x")) for sh in sys.argv[2].split(",")]
    max_lp_pivots = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # stop an LP after about this many pivots (big shapes)
    fused = len(sys.argv) > 4 and sys.argv[4] == "fused"
    kind = oracle.FP64_FUSED if fused else oracle.FP64
    lps.set_default_arithmetic("fused" if fused else "plain")   # (explicit: the library's own default is by size)
    n_lp = n_cmp = pivots_total = 0
    while time.time() < t_end:
        m, n = shapes[n_lp % len(shapes)]
        block = int(rng.choice([8, 16, 32, 32, 64]))
        A = rng.random((m, n))
        b = (n / 4.0) * (1.0 + rng.random(m))
        c = rng.random(n)
        # every third LP with the conservative barrier (release + acquire): the self-check of the acquire-only default
        opts = {"chain_fences": 3} if n_lp % 3 == 2 else {}
        if n_lp % 4 == 1:
            opts["chain_wgs"] = 33     # the decision kernel at full width also on these small shapes
        st = lps.LPState(A, b, c, block=block, options=opts)
        ref = oracle.State(A, b, c, kind=kind)
        lp_piv = 0
        while True:
            budget = int(rng.choice([-1, 1, block, 3 * block + 1, 257, 1000]))
            got = st.simplex_loop(max_pivots=budget)
            want = ref.simplex_loop(max_pivots=budget, threads=8 if m * n > (1 << 21) else 1)
            assert (got[0], got[1]) == (want["status"], want["pivots"]), (n_lp, m, n, block, budget, got, want)
            gA, gb, gc, gv, gp = st.read()
            wA, wb, wc, wv, wp = ref.read()
            ok = (np.array_equal(bits(gA), bits(wA)) and np.array_equal(bits(gb), bits(wb)) and
                  np.array_equal(bits(gc), bits(wc)) and bits(np.array([gv]))[0] == bits(np.array([wv]))[0] and
                  np.array_equal(gp, wp))
            n_cmp += 1
            pivots_total += got[1]
            if not ok:
                print("MISMATCH lp %d shape %dx%d block %d budget %d" % (n_lp, m, n, block, budget))
                sys.exit(1)
            if got[0] != 9:  # not PIVOT_LIMIT: finished
                break
            lp_piv += got[1]
            if max_lp_pivots >= 0 and lp_piv >= max_lp_pivots:
                break
        st.close()
        n_lp += 1
    print("soak ok: %d LPs, %d state comparisons, %d pivots, all bit-identical" % (n_lp, n_cmp, pivots_total))


if __name__ == "__main__":
    main()
