"""Hunt for an intermittent tableau mismatch: one LP, a fixed sequence of budgets through the default loop on fresh
handles, many times; every final tableau compared entry by entry with the fp64 oracle's (computed once).  Prints where
mismatches sit (rows / columns; whether they are pivot rows / entering columns of the run).
    python scripts/flake_hunt.py M N reps budget[,budget...] [name=value options ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

m, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
budgets = [int(x) for x in sys.argv[4].split(",")]
opts = {}
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    opts[k] = int(v)
block = opts.pop("block", None)
rng = np.random.default_rng(3 * m + n)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
ref = oracle.State(A, b, c, kind=oracle.FP64)
for bud in budgets:
    ref.simplex_loop(max_pivots=bud, threads=16)
wA = ref.read()[0]
ref.close()
bad_runs = 0
for rep in range(reps):
    st = lps.LPState(A, b, c, block=block, options=opts)
    for bud in budgets:
        got = st.simplex_loop(max_pivots=bud)
    gA = st.read()[0]
    info = st.info()
    st.close()
    diff = gA.view(np.uint64) != wA.view(np.uint64)
    nbad = int(diff.sum())
    if nbad:
        bad_runs += 1
        rows = np.flatnonzero(diff.any(axis=1))
        cols = np.flatnonzero(diff.any(axis=0))
        print("rep %d: %d entries differ; %d rows (%s ... %s)  %d cols (%s ... %s)  sweep_rows %s" % (
            rep, nbad, rows.size, rows[:12].tolist(), rows[-4:].tolist(), cols.size, cols[:12].tolist(),
            cols[-4:].tolist(), info.get("sweep_rows")), flush=True)
        i, j = np.argwhere(diff)[0]
        print("   first: (%d,%d) got %r want %r" % (i, j, gA[i, j], wA[i, j]), flush=True)
print("%d of %d runs differ (%dx%d budgets %s options %s block %s)" % (bad_runs, reps, m, n, budgets, opts, block))
