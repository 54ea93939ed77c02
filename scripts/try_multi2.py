import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linear_programming_solver_amd as lps
rv = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
case = rv["solve"][2]
form = lps.LPStandardForm(case["A"], case["b"], case["c"], maximize=case["maximize"])
for devs in (None, [0], [0, 0]):
    s = lps.LPSolver(devices=devs)
    try:
        ans = s.solve(form, restore_order=case.get("restore_order"))
    except Exception as ex:
        ans = repr(ex)
    L = s.last
    print(devs, ans, "p1", L.pivots_phase1, "p2", L.pivots_phase2, "x0", L.x0_slot, "obj", L.objective, "perm", list(L.perm), "x", list(L.x))
# phase 1 by hand on LPMulti vs LPState
A = np.array(case["A"], dtype=float); b = np.array(case["b"], dtype=float); m, n = A.shape
auxA = np.hstack([A, -np.ones((m, 1))]); auxc = np.zeros(n + 1); auxc[n] = -1
mib = int(np.argmin(b))
for cls, kw in ((lps.LPState, {}), (lps.LPMulti, {"devices": [0]})):
    st = cls(auxA, b, auxc, **kw)
    st.pivot(n, mib)
    out = st.simplex_loop(track_slot=mib + n + 1)
    Ar, br, cr, vr, pr = st.read()
    print(cls.__name__, out, "v", vr, "b", br, "c", cr, "perm", list(pr)); print(Ar)
print("---- default order")
for devs in (None, [0], [0, 0]):
    s = lps.LPSolver(devices=devs)
    try:
        ans = s.solve(form)
    except Exception as ex:
        ans = repr(ex)
    L = s.last
    print(devs, ans, "p1", L.pivots_phase1, "p2", L.pivots_phase2, "x0", L.x0_slot, "obj", L.objective, "perm", list(map(int, L.perm)), "x", list(L.x))
for order in ([0, 1], [1, 0]):
    for devs in (None, [0]):
        s = lps.LPSolver(devices=devs)
        ans = s.solve(form, restore_order=order)
        print(order, devs, ans, s.last.pivots_phase2)
