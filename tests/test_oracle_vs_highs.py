"""An independent check of what the oracle (and, in the GPU twin of this test, the HIP path) computes: the optimum
of LPSolver.solve against SciPy's HiGHS on seeded random LPs whose start is feasible (b >= 0: no phase 1, so the
reference's restoreInitialLP indexing defect — SURVEY §8a row R9 — cannot play a part), bounded by a sum row or
unbounded by construction.  The reference's rule (first positive reduced cost in, lowest row among equal ratios out)
is Bland's rule, so it terminates; its optimum must be the LP's optimum whatever path it took.
Tolerances: fp64 restatement 1e-9 relative; decimal-15 restatement 1e-9 relative (15 significant digits per operation
over a few hundred pivots)."""
import numpy as np
import pytest

scipy_optimize = pytest.importorskip("scipy.optimize")

TOL = 1e-9


def random_feasible_start_lp(m, n, seed, bounded=True):
    """max c.x  s.t.  A x <= b, x >= 0 with b > 0 (x = 0 is feasible).  bounded: a last row sum(x) <= 10.
    Otherwise column 0 gets c > 0 and no positive entry: the LP is unbounded along x_0."""
    rng = np.random.default_rng(seed)
    A = rng.random((m, n)) * 1.5 - 0.5
    b = 1.0 + rng.random(m)
    c = rng.random(n) * 1.5 - 0.5
    if bounded:
        A[-1, :] = 1.0
        b[-1] = 10.0
    else:
        A[:, 0] = -rng.random(m)
        c[0] = 0.5 + rng.random()
    return A, b, c


def highs(A, b, c):
    res = scipy_optimize.linprog(-c, A_ub=A, b_ub=b, bounds=(0, None), method="highs")
    return res.status, (-res.fun if res.status == 0 else None)


CASES = [(12, 9, 1), (30, 45, 2), (60, 40, 3), (80, 120, 4), (150, 90, 5), (200, 260, 6), (33, 257, 7), (257, 33, 8)]


@pytest.mark.parametrize("m,n,seed", CASES)
def test_oracle_optimum_equals_highs_optimum(oracle, m, n, seed):
    A, b, c = random_feasible_start_lp(m, n, seed)
    status, want = highs(A, b, c)
    assert status == 0
    for kind in (oracle.FP64, oracle.DEC15):
        if kind == oracle.DEC15 and m * n > 20000:
            continue   # the decimal arithmetic is ~1000x slower: small cases only
        res, st = oracle.solve(A, b, c, maximize=True, kind=kind, want_trace=False)
        st.close()
        assert res["status"] == 0 and not res["phase1_used"], (m, n, seed, kind)
        assert abs(res["objective"] - want) <= TOL * max(1.0, abs(want)), (m, n, seed, kind, res["objective"], want)


@pytest.mark.parametrize("m,n,seed", [(10, 8, 11), (40, 60, 12), (120, 70, 13)])
def test_oracle_reports_unbounded_where_highs_does(oracle, m, n, seed):
    A, b, c = random_feasible_start_lp(m, n, seed, bounded=False)
    status, _ = highs(A, b, c)
    assert status == 3     # HiGHS: unbounded
    res, st = oracle.solve(A, b, c, maximize=True, kind=oracle.FP64, want_trace=False)
    st.close()
    assert res["status"] == 1   # LPX_UNBOUNDED: SolutionException("This linear program is unbounded"), LPSolver.java:105


def test_minimisation_goes_through_the_same_path(oracle):
    """LPSolver.solve(minimize) negates c and the objective (LPSolver.java:81-93)."""
    A, b, c = random_feasible_start_lp(50, 70, 21)
    res_h = scipy_optimize.linprog(c, A_ub=A, b_ub=b, bounds=(0, None), method="highs")
    assert res_h.status == 0
    res, st = oracle.solve(A, b, c, maximize=False, kind=oracle.FP64, want_trace=False)
    st.close()
    assert res["status"] == 0
    assert abs(res["objective"] - res_h.fun) <= TOL * max(1.0, abs(res_h.fun))


def random_infeasible_lp(m, n, seed):
    """A random LP with b of mixed sign (phase 1 runs) made infeasible by one contradictory pair of rows:
    w.x <= 1 and -w.x <= -3 with w > 0.  The verdict comes out of phase 1 alone (x0 > 0 at the auxiliary optimum,
    LPSolver.java:168-173), before restoreInitialLP is ever reached."""
    rng = np.random.default_rng(seed)
    A = rng.random((m, n)) * 1.5 - 0.5
    b = rng.random(m) * 3.0 - 1.0
    c = rng.random(n)
    w = 0.5 + rng.random(n)
    A[0, :], b[0] = w, 1.0
    A[1, :], b[1] = -w, -3.0
    return A, b, c


@pytest.mark.parametrize("m,n,seed", [(6, 5, 31), (25, 40, 32), (90, 60, 33)])
def test_oracle_reports_infeasible_where_highs_does(oracle, m, n, seed):
    A, b, c = random_infeasible_lp(m, n, seed)
    assert highs(A, b, c)[0] == 2      # HiGHS: infeasible
    for kind in (oracle.FP64, oracle.DEC15):
        res, st = oracle.solve(A, b, c, maximize=True, kind=kind, want_trace=False)
        st.close()
        assert res["phase1_used"] and res["status"] == 2, (m, n, seed, kind, res["status"])   # LPX_INFEASIBLE
