#!/usr/bin/env python3
"""Divergence census: how often does each BINARY arithmetic mode leave the decimal-15 pivot sequence?

The reference computes in BigDecimal with MathContext(15, HALF_UP) (LPState.java:18): two decimal roundings per
update (`:162`).  The engine offers two binary modes: "plain" (IEEE fp64, product and difference rounded separately) and
"fused" (one v_fma_f64 per update).  Neither can reproduce decimal roundings; what north_star asks for is the same basis
and the objective within tolerance (the reference itself keeps 6 decimals, LPSolver.java:113).  Which mode is the
default must follow from a measurement, not from taste (Weak 7 / Next 3): this script solves every LP of
four seeded families with the three oracle instantiations (oracle/lp_oracle.hpp: Dec15, double, F64Fused) and records, per
LP and binary mode, the index of the first pivot that differs from the decimal-15 sequence (-1: none), whether the final
basis and the 6-decimal text agree, and the relative objective difference.

    python tests/golden/gen_divergence_census.py            # ~10 min on 8 cores -> tests/golden/divergence_census.json
    CENSUS_QUICK=1 python tests/golden/gen_divergence_census.py   # the small sizes only (what the CPU test re-runs)

Families (NumPy default_rng(seed); sizes m x n):
  dense_u01      A ~ U(0,1), b = (n/4) U(1,2), c ~ U(0,1)            (BASELINE cfg2-cfg4's generator)
  dense_6dec     the same rounded to 6 decimals: inputs exact in decimal, inexact in binary
  packing_01     A in {0,1} (density 0.15), integer b in 1..4, integer c in 1..3: massive ratio ties; quotients by 3, 7, ...
                 are exact in NEITHER radix, so ties may break differently — the family most likely to diverge
  degenerate_int tests/golden/gen_cfg5.make_cfg5 at small sizes: phase 1 + ties, entries stay dyadic (cfg5's family)
"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "divergence_census.json")


def make_lp(family, m, n, seed):
    rng = np.random.default_rng(seed)
    if family in ("dense_u01", "dense_6dec"):
        A = rng.random((m, n))
        b = (n / 4.0) * (1.0 + rng.random(m))
        c = rng.random(n)
        if family == "dense_6dec":
            A, b, c = np.round(A, 6), np.round(b, 6), np.round(c, 6)
        return A, b, c
    if family == "packing_01":
        A = (rng.random((m, n)) < 0.15).astype(np.float64)
        for j in range(n):   # every variable in at least one row: bounded
            if not A[:, j].any():
                A[int(rng.integers(0, m)), j] = 1.0
        b = rng.integers(1, 5, size=m).astype(np.float64)
        c = rng.integers(1, 4, size=n).astype(np.float64)
        return A, b, c
    if family == "degenerate_int":
        from tests.golden.gen_cfg5 import make_cfg5
        return make_cfg5(m, n, seed)
    raise ValueError(family)


def plan(quick):
    cases = []
    small = [("dense_u01", 64, 128, 60), ("dense_6dec", 64, 128, 60), ("packing_01", 48, 96, 60), ("packing_01", 96, 160, 40),
             ("degenerate_int", 60, 60, 30), ("degenerate_int", 120, 120, 20)]
    big = [("dense_u01", 128, 256, 40), ("dense_6dec", 128, 256, 40), ("dense_u01", 256, 512, 16), ("dense_6dec", 256, 512, 16),
           ("packing_01", 192, 320, 16), ("degenerate_int", 240, 240, 10)]
    for fam, m, n, count in small + ([] if quick else big):
        for s in range(1, count + 1):
            cases.append((fam, m, n, s))
    return cases


def first_divergence(ref, got):
    k = min(len(ref), len(got))
    d = np.nonzero((ref[:k] != got[:k]).any(axis=1))[0]
    if d.size:
        return int(d[0])
    return -1 if len(ref) == len(got) else k


def run_case(case):
    from oracle import pyoracle as orc
    fam, m, n, seed = case
    A, b, c = make_lp(fam, m, n, seed)
    rd, sd = orc.solve(A, b, c, True, kind=orc.DEC15, threads=1)
    td = rd["trace"]
    perm_d = sd.read()[4]
    rec = {"family": fam, "m": m, "n": n, "seed": seed, "status": int(rd["status"]), "pivots": int(rd["pivots1"] + rd["pivots2"]),
           "phase1": bool(rd["phase1_used"]), "text": rd["objective_text"]}
    sd.close()
    for name, kind in (("plain", orc.FP64), ("fused", orc.FP64_FUSED)):
        r, s = orc.solve(A, b, c, True, kind=kind, threads=1)
        perm = s.read()[4]
        s.close()
        dv = first_divergence(td, r["trace"])
        same_basis = bool(perm is not None and perm_d is not None and len(perm) == len(perm_d) and (perm == perm_d).all())
        od, ob = rd["objective"], r["objective"]
        rel = abs(od - ob) / max(1.0, abs(od)) if r["status"] == rd["status"] == 0 else None
        rec[name] = {"first_divergence": dv, "status": int(r["status"]), "pivots": int(r["pivots1"] + r["pivots2"]),
                     "same_final_basis": same_basis, "same_text": r["objective_text"] == rd["objective_text"],
                     "objective_rel_diff": rel}
    return rec


def summarise(records):
    out = {}
    for r in records:
        key = "%s %dx%d" % (r["family"], r["m"], r["n"])
        g = out.setdefault(key, {"lps": 0, "pivots": 0,
                                 "plain": {"diverged": 0, "basis_differs": 0, "text_differs": 0, "status_differs": 0},
                                 "fused": {"diverged": 0, "basis_differs": 0, "text_differs": 0, "status_differs": 0}})
        g["lps"] += 1
        g["pivots"] += r["pivots"]
        dp, df = r["plain"]["first_divergence"], r["fused"]["first_divergence"]
        big = 1 << 62
        g["fused_leaves_earlier"] = g.get("fused_leaves_earlier", 0) + ((df if df >= 0 else big) < (dp if dp >= 0 else big))
        g["plain_leaves_earlier"] = g.get("plain_leaves_earlier", 0) + ((dp if dp >= 0 else big) < (df if df >= 0 else big))
        for mode in ("plain", "fused"):
            x = r[mode]
            g[mode]["diverged"] += x["first_divergence"] != -1
            g[mode]["basis_differs"] += not x["same_final_basis"]
            g[mode]["text_differs"] += not x["same_text"]
            g[mode]["status_differs"] += x["status"] != r["status"]
    total = {"lps": 0, "pivots": 0, "fused_leaves_earlier": 0, "plain_leaves_earlier": 0, "plain": {}, "fused": {}}
    for g in out.values():
        for k in ("lps", "pivots", "fused_leaves_earlier", "plain_leaves_earlier"):
            total[k] += g[k]
        for mode in ("plain", "fused"):
            for k, v in g[mode].items():
                total[mode][k] = total[mode].get(k, 0) + v
    out["TOTAL"] = total
    return out


def main():
    quick = bool(int(os.environ.get("CENSUS_QUICK", "0")))
    cases = plan(quick)
    t = time.time()
    with Pool(int(os.environ.get("CENSUS_PROCS", "8"))) as pool:
        records = []
        for i, rec in enumerate(pool.imap(run_case, cases, chunksize=1)):
            records.append(rec)
            if (i + 1) % 25 == 0:
                print("%d / %d  (%.0f s)" % (i + 1, len(cases), time.time() - t), flush=True)
    summ = summarise(records)
    doc = {"generator": "tests/golden/gen_divergence_census.py", "quick": quick,
           "oracle": "oracle/lp_oracle.hpp: Dec15 (reference semantics) vs double (plain) vs F64Fused (fused)",
           "first_divergence": "index of the first pivot (phase, entering, leaving) that differs from the decimal-15 sequence; -1 = none",
           "summary": summ, "records": records}
    if not quick:
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=None, separators=(",", ":"))
            f.write("\n")
    for k, g in summ.items():
        print("%-28s %4d LPs %8d pivots | plain: %s | fused: %s | leaves the decimal sequence earlier: plain %d, fused %d" % (
            k, g["lps"], g["pivots"], g["plain"], g["fused"], g["plain_leaves_earlier"], g["fused_leaves_earlier"]))
    print("%.0f s" % (time.time() - t))


if __name__ == "__main__":
    main()
