import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps
rv = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
for rep in range(2):
    for ci, case in enumerate(rv["solve"]):
        m = len(case["b"])
        for ndev in (1, 2, 3):
            if ndev > m:
                continue
            form = lps.LPStandardForm(case["A"], case["b"], case["c"], maximize=case["maximize"])
            s = lps.LPSolver(devices=[0] * ndev)
            try:
                ans = s.solve(form, restore_order=case.get("restore_order"))
            except Exception as ex:
                ans = type(ex).__name__
            L = s.last
            print(rep, ci, ndev, case["status"], case.get("answer"), "->", ans, "p1", L.pivots_phase1, "p2", L.pivots_phase2, "x0", L.x0_slot, "obj", L.objective, flush=True)
