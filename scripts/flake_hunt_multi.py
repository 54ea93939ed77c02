"""Hunt for an intermittent failure of the multi-device path rehearsed on ONE GPU (round 4 saw a four-shard rehearsal trip its
spin bound once in seven runs): one LP, a fixed budget sequence through lpx_multi on FRESH handles with `shards` row blocks,
all on device 0, many times; every run's status, pivot count, objective and basis compared with the fp64 oracle's (computed
once).  Reports every run that differs or ends in LPX_DEVICE_ERROR (7), with the engine's reserved word.
    GPU_MAX_HW_QUEUES=16 python scripts/flake_hunt_multi.py M N shards reps budget[,budget...] [name=value options ...]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

m, n, shards, reps = (int(x) for x in sys.argv[1:5])
budgets = [int(x) for x in sys.argv[5].split(",")]
opts = {}
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    opts[k] = int(v)
block = opts.pop("block", None)
lps.set_default_arithmetic("plain")
rng = np.random.default_rng(5 * m + n)
A = rng.random((m, n))
b = (n / 4.0) * (1.0 + rng.random(m))
c = rng.random(n)
ref = oracle.State(A, b, c, kind=oracle.FP64)
want = [ref.simplex_loop(max_pivots=bud, threads=16) for bud in budgets]
_, wb, wc, wv, wperm = ref.read()
ref.close()
bad = []
t0 = time.time()
for rep in range(reps):
    mt = lps.LPMulti(A, b, c, devices=[0] * shards, block=block, options=opts)
    got = [mt.simplex_loop(max_pivots=bud)[:2] for bud in budgets]
    ok = all(g == (w["status"], w["pivots"]) for g, w in zip(got, want))
    if ok:
        _, gb, gc, gv, gperm = mt.read(want_A=False)
        ok = gv == wv and list(gperm) == list(wperm) and np.array_equal(np.asarray(gb).view(np.uint64), wb.view(np.uint64))
    if not ok:
        bad.append((rep, got))
        print("run %d differs: %s (wanted %s)" % (rep, got, [(w["status"], w["pivots"]) for w in want]), flush=True)
    mt.close()
    if (rep + 1) % 100 == 0:
        print("%d runs, %d bad, %.0f s" % (rep + 1, len(bad), time.time() - t0), flush=True)
print("flake hunt %dx%d, %d shards on one GPU, budgets %s, options %s: %d of %d runs differ" % (m, n, shards, budgets, opts, len(bad), reps))
