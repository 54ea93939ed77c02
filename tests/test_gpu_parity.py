"""GPU parity tests proper: the HIP path, called through the C ABI (liblpx.so via the host mirror classes),
against the oracle on the same inputs.

Bars: bit-exact against the fp64 instantiation of the oracle (every tableau entry, b, c, v, the slot
permutation, pivot counts and statuses); against the decimal-15 instantiation (the reference's BigDecimal
semantics) identical pivot sequence / basis and objective within OBJ_TOL.  The reference's own Spock vectors
are replayed through the device as well."""
from decimal import Decimal

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OBJ_TOL = 1e-9           # |v_gpu - v_decimal| <= OBJ_TOL * max(1, |v|)   (north_star tolerance)
STATUS = {"OPTIMAL": 0, "UNBOUNDED": 1, "INFEASIBLE": 2}


@pytest.fixture(scope="module")
def lps(arith):
    """The host package; every test of this module runs in both arithmetic modes (tests/conftest.py `arith`)."""
    from tests.conftest import package_in_mode
    pkg = package_in_mode(arith)
    yield pkg
    pkg.set_default_arithmetic("auto")


@pytest.fixture(scope="module")
def oracle(arith):
    """The checker of the current mode: oracle.FP64 is the fp64 instantiation ("plain") or the fused one ("fused")."""
    from oracle import pyoracle
    from tests.conftest import ArithOracle
    pyoracle.build()
    pyoracle.lib()
    return ArithOracle(pyoracle, arith)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_state_bits_equal(got, want, what=""):
    gA, gb, gc, gv, gp = got
    wA, wb, wc, wv, wp = want
    assert np.array_equal(bits(gA), bits(wA)), "A differs " + what
    assert np.array_equal(bits(gb), bits(wb)), "b differs " + what
    assert np.array_equal(bits(gc), bits(wc)), "c differs " + what
    assert bits(np.array([gv]))[0] == bits(np.array([wv]))[0], "v differs %s: %r vs %r" % (what, gv, wv)
    if wp is not None:
        assert list(gp) == list(wp), "perm differs " + what


def dense_lp(m, n, seed):
    """SURVEY §8(d) synthetic input: A ~ U(0,1), b = (n/4) U(1,2), c ~ U(0,1), maximise."""
    rng = np.random.default_rng(seed)
    return rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)


# ------------------------------------------------------------------------------------ reference vectors
def test_reference_get_entering(lps, reference_vectors):          # LPStateSpec.groovy:12-29
    for case in reference_vectors["get_entering"]["cases"]:
        c = case["c"]
        st = lps.LPState(np.zeros((0, len(c))), [], c)
        assert st.get_entering() == case["entering"], case
        st.close()


def test_reference_get_leaving(lps, reference_vectors):           # LPStateSpec.groovy:31-48
    g = reference_vectors["get_leaving"]
    st = lps.LPState(g["A"], g["b"], [0, 0, 0, 0])
    for case in g["cases"]:
        assert st.get_leaving(case["entering"]) == case["leaving"], case
    with pytest.raises(ValueError):                                # Validate.isTrue -> IllegalArgumentException
        st.get_leaving(4)
    with pytest.raises(ValueError):
        st.get_leaving(-1)
    st.close()


def test_reference_pivot_vectors(lps, reference_vectors):         # LPStateSpec.groovy:50-163
    for group in reference_vectors["pivot"]:
        for case in group["cases"]:
            n, m = len(group["c"]), len(group["b"])
            names = {s: "x%d" % (i + 1) for s, i in enumerate(group["perm"])}
            st = lps.LPState(group["A"], group["b"], group["c"], variables=names,
                             coefficients={v: k for k, v in names.items()})
            st.pivot(case["entering"], case["leaving"])
            A, b, c, v, perm = st.read()
            assert np.array_equal(A, np.array(case["resA"], dtype=float)), (group["source"], case)
            assert np.array_equal(b, np.array(case["resB"], dtype=float))
            assert np.array_equal(c, np.array(case["resC"], dtype=float))
            assert v == case["resV"]
            assert list(perm) == case["resPerm"]
            # the reference's name maps after exchangeIndexes (LPState.java:311-320)
            assert st.variables == {s: "x%d" % (i + 1) for s, i in enumerate(case["resPerm"])}
            assert st.coefficients == {"x%d" % (i + 1): s for s, i in enumerate(case["resPerm"])}
            st.close()


def test_reference_solve_vectors(lps, reference_vectors):         # LPSolverSpec.groovy:76-111, :151-192
    for case in reference_vectors["solve"]:
        form = lps.LPStandardForm(case["A"], case["b"], case["c"], maximize=case["maximize"])
        solver = lps.LPSolver()
        if case["status"] == "OPTIMAL":
            ans = solver.solve(form, restore_order=case.get("restore_order"))
            assert ans == Decimal(case["answer"]), (case["source"], ans)
        else:
            exc = lps.SolutionException if case["status"] == "UNBOUNDED" else lps.LPException
            with pytest.raises(exc) as ei:
                solver.solve(form)
            assert str(ei.value) == case["message"], case["source"]
            assert solver.last.status == STATUS[case["status"]]


def test_reference_initial_infeasible_x0_slot(lps, reference_vectors):   # logs/lp_solver.log:196
    case = reference_vectors["solve"][2]
    solver = lps.LPSolver()
    solver.solve(lps.LPStandardForm(case["A"], case["b"], case["c"], maximize=True), restore_order=[0, 1])
    assert solver.last.phase1_used and solver.last.x0_slot == 1


def test_io_files_first_block_end_to_end(lps, reference_vectors):        # cfg1: io_files/input.txt:1-16 -> 7.000000
    g = reference_vectors["io_files_first_block"]
    form = lps.LPInputReader().read_lp(g["text"])
    assert (form.m, form.n) == (g["m"], g["n"])
    solver = lps.LPSolver()
    assert str(solver.solve(form)) == g["objective_text"]
    x = solver.last.x
    assert np.all(x >= -1e-12) and np.all(form.A @ x <= form.b + 1e-9) and abs(x.sum() - 7.0) < 1e-9


def test_zero_pivot_is_divide_by_zero(lps):
    st = lps.LPState([[0.0, 1.0], [1.0, 1.0]], [1.0, 2.0], [1.0, 1.0])
    with pytest.raises(ZeroDivisionError):
        st.pivot(0, 0)
    A, b, c, v, perm = st.read()
    assert np.array_equal(A, [[0.0, 1.0], [1.0, 1.0]]) and list(perm) == [0, 1, 2, 3]   # untouched
    st.close()


# ------------------------------------------------------------------------------------ golden LP cases
def test_golden_cases_bit_exact_vs_fp64_oracle_and_decimal_tolerance(lps, oracle, decimal_goldens):
    from linear_programming_solver_amd import _lib
    for case in decimal_goldens["lp_cases"]:
        m, n = case["m"], case["n"]
        A = np.array([float(x) for x in case["A"]]).reshape(m, n)
        b = np.array([float(x) for x in case["b"]])
        c = np.array([float(x) for x in case["c"]])
        want, wst = oracle.solve(A, b, c, maximize=case["maximize"], kind=oracle.FP64)
        solver = lps.LPSolver()
        try:
            solver.solve(lps.LPStandardForm(A, b, c, maximize=case["maximize"]))
        except (lps.LPException, IndexError):
            pass
        got = solver.last
        name = case["name"]
        assert got.status == want["status"], name
        assert got.phase1_used == want["phase1_used"], name
        assert (got.pivots_phase1, got.pivots_phase2) == (want["pivots1"], want["pivots2"]), name
        assert got.x0_slot == want["x0_slot"], name
        assert bits(np.array([got.objective]))[0] == bits(np.array([want["objective"]]))[0], name
        if got.status == 0:
            assert list(got.perm) == list(wst.read()[4]), name
            assert got.objective_text == want["objective_text"], name
        # against the decimal-15 (reference-semantics) golden: same status and pivot counts; objective in tol
        if want["trace"].tolist() == case["trace"]:
            assert got.status == case["status"], name
            if case["status"] == 0:
                ref = float(Decimal(case["objective_repr"]))
                assert abs(got.objective - ref) <= OBJ_TOL * max(1.0, abs(ref)), name
                assert got.objective_text == case["objective_text"], name


def test_final_tableau_bits_after_solve(lps, oracle):
    """Whole-solve parity of every tableau entry (kept state handle), feasible and infeasible start."""
    import ctypes as C
    from linear_programming_solver_amd import _lib
    rng = np.random.default_rng(11)
    for trial in range(4):
        m, n = [(12, 20), (33, 17), (40, 64), (25, 25)][trial]
        A = rng.integers(-4, 9, size=(m, n)).astype(float) + rng.integers(0, 4, size=(m, n)) / 4.0
        b = rng.integers(-3 if trial % 2 else 1, 40, size=m).astype(float)
        c = rng.integers(-2, 6, size=n).astype(float)
        want, wst = oracle.solve(A, b, c, True, kind=oracle.FP64)
        L = _lib.lib()
        opts = _lib.SolveOptions()
        opts.fused = 1 if _lib.DEFAULT_FUSED else 0     # the raw C call: the arithmetic mode this test runs in
        opts.max_pivots = -1
        keep = C.c_void_p()
        opts.keep_state = C.pointer(keep)
        res = _lib.SolveResult()
        L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp), n, b.ctypes.data_as(_lib.dp), c.ctypes.data_as(_lib.dp), 1,
                    C.byref(opts), C.byref(res))
        assert res.status == want["status"], trial
        assert (res.pivots_phase1, res.pivots_phase2) == (want["pivots1"], want["pivots2"])
        assert keep.value
        fm, fn = C.c_int32(), C.c_int32()
        L.lpx_state_dims(keep, C.byref(fm), C.byref(fn), None, None)
        gA = np.zeros((fm.value, fn.value)); gb = np.zeros(fm.value); gc = np.zeros(fn.value)
        gv = C.c_double(); gp = np.zeros(fm.value + fn.value, dtype=np.int32)
        L.lpx_state_read(keep, gA.ctypes.data_as(_lib.dp), fn.value, gb.ctypes.data_as(_lib.dp),
                         gc.ctypes.data_as(_lib.dp), C.byref(gv), gp.ctypes.data_as(_lib.ip))
        L.lpx_state_destroy(keep)
        assert (fm.value, fn.value) == (wst.m, wst.n)
        assert_state_bits_equal((gA, gb, gc, gv.value, gp), wst.read(), "trial %d" % trial)


# ------------------------------------------------------------------------------------ step-level parity
@pytest.mark.parametrize("shape", [(7, 5), (64, 100), (130, 513), (300, 2100), (96, 9000)])
def test_step_api_every_pivot_bit_exact(lps, oracle, shape):
    """getEntering / getLeaving / pivot one at a time, full state compared after every pivot.  Shapes cover
    one strip / several strips / ragged tails of the k_update tiling (n not a multiple of 16, odd m)."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=m * 1000 + n)
    st = lps.LPState(A, b, c)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for k in range(6):
        e = st.get_entering()
        assert e == ref.get_entering()
        if e < 0:
            break
        l = st.get_leaving(e)
        assert l == ref.get_leaving(e)
        st.pivot(e, l)
        assert ref.pivot(e, l) == 0
        assert_state_bits_equal(st.read(), ref.read(), "after pivot %d of %s" % (k, shape))
    st.close()


@pytest.mark.parametrize("shape", [(50, 80), (257, 300), (200, 1100)])
def test_device_loop_matches_step_api_and_oracle(lps, oracle, shape):
    """The fused device-resident loop (select_pivot + update with by-product ratio test) against the oracle
    after a fixed pivot budget and at optimality."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=7 * m + n)
    for budget in (1, 2, 17, -1):
        st = lps.LPState(A, b, c)
        ref = oracle.State(A, b, c, kind=oracle.FP64)
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, budget)
        assert_state_bits_equal(st.read(), ref.read(), "budget %d of %s" % (budget, shape))
        st.close()


def test_loop_can_be_resumed(lps, oracle):
    A, b, c = dense_lp(80, 120, seed=3)
    st = lps.LPState(A, b, c)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    total = 0
    for chunk in (3, 5, 1, 40, -1):
        status, pivots, _ = st.simplex_loop(max_pivots=chunk)
        want = ref.simplex_loop(max_pivots=chunk)
        assert (status, pivots) == (want["status"], want["pivots"])
        total += pivots
        assert_state_bits_equal(st.read(), ref.read(), "after chunk %d" % chunk)
    assert status == 0 and total > 40
    st.close()


def test_degenerate_ties_lowest_row_wins(lps, oracle):
    """Equal ratios everywhere: the leaving row must be the lowest index (strict '<' scan, LPState.java:299)."""
    m, n = 70, 40
    A = np.ones((m, n)); b = np.full(m, 3.0); c = np.arange(n, 0, -1).astype(float)
    st = lps.LPState(A, b, c)
    assert st.get_entering() == 0 and st.get_leaving(0) == 0
    status, pivots, _ = st.simplex_loop()
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop()
    assert (status, pivots) == (want["status"], want["pivots"])
    assert_state_bits_equal(st.read(), ref.read())
    st.close()


def test_unbounded_in_device_loop(lps):
    st = lps.LPState([[1.0, 0.0]], [1.0], [1.0, 1.0])              # LPSolverSpec.groovy:151-163
    status, pivots, _ = st.simplex_loop()
    assert (status, pivots) == (1, 1)
    st.close()


def test_empty_and_tiny_shapes(lps, oracle):
    st = lps.LPState(np.zeros((0, 3)), [], [0.0, -1.0, 0.0])
    assert st.get_entering() == -1
    assert st.simplex_loop()[:2] == (0, 0)
    st.close()
    st = lps.LPState([[2.0]], [4.0], [3.0])
    assert st.simplex_loop()[:2] == (0, 1) and st.v == 6.0
    st.close()


# ------------------------------------------------------------------------------------ cfg2: full solve
def test_cfg2_full_solve_matches_fp64_oracle_bit_for_bit(lps, oracle):
    """BASELINE cfg2: dense random LP m=1024 n=2048, full solve.  ~22.7k pivots; the fp64 oracle pins the
    pivot count, the final basis permutation, the objective bits and position-keyed checksums of A, b, c."""
    from linear_programming_solver_amd.lp_state import checksum_host
    m, n = 1024, 2048
    A, b, c = dense_lp(m, n, seed=1)
    st = lps.LPState(A, b, c)
    status, pivots, _ = st.simplex_loop()
    want, wst = oracle.solve(A, b, c, True, kind=oracle.FP64, threads=8, want_trace=False)
    assert status == 0 and want["status"] == 0
    assert pivots == want["pivots2"]
    wA, wb, wc, wv, wperm = wst.read()
    assert st.checksum() == checksum_host(wA, wb, wc)
    _, gb, gc, gv, gperm = st.read(want_A=False)
    assert list(gperm) == list(wperm)
    assert bits(np.array([gv]))[0] == bits(np.array([wv]))[0]
    assert np.array_equal(bits(gb), bits(wb)) and np.array_equal(bits(gc), bits(wc))
    st.close()


@pytest.mark.parametrize("m,n,seed", [(12, 9, 1), (80, 120, 4), (200, 260, 6), (33, 257, 7), (257, 33, 8), (600, 900, 9)])
def test_solve_optimum_equals_highs_optimum(lps, m, n, seed):
    """Independent of the oracle: lpx_solve's optimum, its solution vector and its unbounded verdict against SciPy's
    HiGHS on seeded random LPs with a feasible start (tests/test_oracle_vs_highs.py holds the generator and the
    reasoning); 1e-9 relative on the objective, x* feasible and reaching the same objective."""
    from tests.test_oracle_vs_highs import TOL, highs, random_feasible_start_lp
    A, b, c = random_feasible_start_lp(m, n, seed)
    status, want = highs(A, b, c)
    assert status == 0
    solver = lps.LPSolver()
    solver.solve(lps.LPStandardForm(A, b, c, maximize=True))
    got = solver.last
    assert got.status == 0 and not got.phase1_used
    assert abs(got.objective - want) <= TOL * max(1.0, abs(want)), (got.objective, want)
    x = got.x
    assert np.all(x >= 0.0) and np.all(A @ x <= b + 1e-9 * (1.0 + np.abs(b)))
    assert abs(float(c @ x) - want) <= 1e-8 * max(1.0, abs(want))
    Au, bu, cu = random_feasible_start_lp(m, n, seed + 100, bounded=False)
    assert highs(Au, bu, cu)[0] == 3
    with pytest.raises(lps.SolutionException):       # "This linear program is unbounded", LPSolver.java:105
        lps.LPSolver().solve(lps.LPStandardForm(Au, bu, cu, maximize=True))


@pytest.mark.parametrize("m,n,seed", [(6, 5, 31), (25, 40, 32), (90, 60, 33), (300, 500, 34)])
def test_solve_reports_infeasible_where_highs_does(lps, m, n, seed):
    """Phase 1 on the device against an independent solver: a random LP with a contradictory pair of rows must end
    in LPException("This linear program is infeasible") (LPSolver.java:171-174), on one GPU and on row-block shards."""
    from tests.test_oracle_vs_highs import highs, random_infeasible_lp
    A, b, c = random_infeasible_lp(m, n, seed)
    assert highs(A, b, c)[0] == 2
    for devices in (None, [0, 0]):
        solver = lps.LPSolver(devices=devices)
        with pytest.raises(lps.LPException) as err:
            solver.solve(lps.LPStandardForm(A, b, c, maximize=True))
        assert not isinstance(err.value, lps.SolutionException)
        assert solver.last.status == 2 and solver.last.phase1_used


def test_cfg2_first_pivots_follow_the_decimal_reference(lps, oracle):
    """At cfg2 the decimal-15 oracle (reference arithmetic) is too slow for a full solve; it pins the first
    pivots: same (entering, leaving) sequence, objective within OBJ_TOL."""
    m, n = 1024, 2048
    A, b, c = dense_lp(m, n, seed=1)
    A = np.round(A, 6); b = np.round(b, 6); c = np.round(c, 6)       # <= 15 significant digits: exact decimals
    K = 6
    dec = oracle.State(A, b, c, kind=oracle.DEC15)
    want = dec.simplex_loop(max_pivots=K, threads=8, want_trace=True)
    f64 = oracle.State(A, b, c, kind=oracle.FP64)
    w64 = f64.simplex_loop(max_pivots=K, threads=8, want_trace=True)
    assert want["trace"].tolist() == w64["trace"].tolist()
    st = lps.LPState(A, b, c)
    status, pivots, _ = st.simplex_loop(max_pivots=K)
    assert pivots == K
    assert_state_bits_equal(st.read(), f64.read())
    dv = dec.read()[3]
    assert abs(st.v - dv) <= OBJ_TOL * max(1.0, abs(dv))
    assert list(st.perm) == list(dec.read()[4])
    st.close()


def test_cfg2_full_solve_vs_the_decimal_reference_golden(lps):
    """BASELINE cfg2's own criterion at full length: "objective within 1e-9 of the Java reference".  The decimal-15
    oracle (BigDecimal / MathContext(15, HALF_UP) semantics, LPState.java:18) solved the 6-decimal cfg2 instance to
    optimality once (22 704 pivots, 13 minutes of CPU: tests/golden/gen_cfg2.py -> cfg2_golden_1024x2048.json); the
    device must reach the same basis in the same number of pivots, the unrounded objective within OBJ_TOL and the same
    6-decimal text (LPSolver.java:113).  If binary and decimal rounding ever ordered a near-tie differently the golden
    names the first diverging pivot (none: -1) and the basis assertion below would say so."""
    import json
    import os
    from tests.golden.gen_cfg2 import make_cfg2
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg2_golden_1024x2048.json")))
    A, b, c = make_cfg2(g["m"], g["n"], g["seed"])
    solver = lps.LPSolver()
    ans = solver.solve(lps.LPStandardForm(A, b, c, maximize=True))
    got = solver.last
    assert got.status == 0 and not got.phase1_used
    assert got.pivots_phase2 == g["pivots"], (got.pivots_phase2, g["pivots"], "first divergence", g["first_divergence"])
    assert got.perm.tolist() == g["perm"], "final basis differs from the decimal reference's"
    want = g["objective_float"]
    assert abs(got.objective - want) <= OBJ_TOL * max(1.0, abs(want)), (got.objective, g["objective_repr"])
    assert str(ans) == g["objective_text"]


# ------------------------------------------------------------------------------------ row-block shards
@pytest.mark.parametrize("lookahead,pipeline", [(False, 1), (True, 1), (True, 2)])
@pytest.mark.parametrize("nshards,shape,budget", [(2, (64, 100), -1), (4, (130, 513), 25), (8, (257, 2100), 12),
                                                  (3, (10, 40), -1), (1, (300, 700), -1), (5, (1000, 260), 40),
                                                  (2, (90, 700), 1), (2, (90, 700), 2), (2, (33, 64), 0)])
def test_shard_kernels_on_one_gpu_match_oracle(lps, oracle, nshards, shape, budget, lookahead, pipeline):
    """k_propose / k_commit / k_update on row-block shards, all shards living on this one GPU and exchanging
    through LocalExchange (the multi-GPU protocol minus RCCL): bit-exact against the unsharded oracle."""
    import torch
    from linear_programming_solver_amd.sharded import HipShardEngine, LocalExchange, row_block, sharded_simplex_loop
    m, n = shape
    A, b, c = dense_lp(m, n, seed=m + n)
    stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
    engines = []
    for r in range(nshards):
        r0, r1 = row_block(m, nshards, r)
        engines.append(HipShardEngine(A[r0:r1], b[r0:r1], c, r0, m, nshards, device=0, stream=stream,
                                      comm_stream=comm, pipeline=pipeline))
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    if budget >= 0 and budget <= 2:
        # resumed runs: odd/even pivot counts leave the tableau in either buffer of the out-of-place pipeline
        for _ in range(3):
            status, pivots, _ = sharded_simplex_loop(engines, LocalExchange(), max_pivots=budget, poll_every=7,
                                                     lookahead=lookahead)
            want = ref.simplex_loop(max_pivots=budget)
            assert (status, pivots) == (want["status"], want["pivots"])
    status, pivots, _ = sharded_simplex_loop(engines, LocalExchange(), max_pivots=budget, poll_every=7,
                                             lookahead=lookahead)
    want = ref.simplex_loop(max_pivots=budget)
    assert (status, pivots) == (want["status"], want["pivots"])
    wA, wb, wc, wv, wperm = ref.read()
    for e in engines:
        gA, gb, gc, gv, gperm = e.read()
        r0 = e.row0
        assert np.array_equal(bits(gA), bits(wA[r0:r0 + e.m_local]))
        assert np.array_equal(bits(gb), bits(wb[r0:r0 + e.m_local]))
        assert np.array_equal(bits(gc), bits(wc)) and gv == wv and list(gperm) == list(wperm)
        e.close()


# ------------------------------------------------------------------------------------ cfg5: phase 1, degenerate
def test_cfg5_degenerate_phase1_basis_bit_exact_vs_reference_semantics(lps, oracle):
    """BASELINE cfg5: m = n = 4096 integer LP with negative right-hand sides (auxiliary-LP phase 1, forced
    first pivot, x0 tracking, column drop + objective restore) and massive ratio ties.  The basis permutation,
    pivot counts, x0's final slot and the objective text must equal the committed decimal-15 golden (the
    reference's BigDecimal semantics, tests/golden/gen_cfg5.py) exactly, and the whole final state must equal
    the fp64 oracle bit for bit."""
    import ctypes as C
    import json
    import os
    from linear_programming_solver_amd import _lib
    from linear_programming_solver_amd.lp_state import checksum_host
    from tests.golden.gen_cfg5 import make_cfg5
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg5_golden_4096x4096.json")))
    m, n = gold["m"], gold["n"]
    A, b, c = make_cfg5(m, n, gold["seed"])
    L = _lib.lib()
    opts = _lib.SolveOptions()
    opts.fused = 1 if _lib.DEFAULT_FUSED else 0     # the raw C call: the arithmetic mode this test runs in
    opts.max_pivots = -1
    keep = C.c_void_p()
    opts.keep_state = C.pointer(keep)
    perm = np.zeros(n + m, dtype=np.int32)
    opts.perm_out = perm.ctypes.data_as(_lib.ip)
    res = _lib.SolveResult()
    rc = L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp), n, b.ctypes.data_as(_lib.dp), c.ctypes.data_as(_lib.dp), 1,
                     C.byref(opts), C.byref(res))
    assert rc == 0 and res.status == gold["status"] == 0
    assert res.phase1_used == 1
    assert (res.pivots_phase1, res.pivots_phase2) == (gold["pivots_phase1"], gold["pivots_phase2"])
    assert res.x0_slot == gold["x0_slot"]
    assert res.objective_text.decode() == gold["objective_text"]
    first_diff = next((i for i in range(n + m) if perm[i] != gold["perm"][i]), None)
    assert first_diff is None, "basis permutation diverges from the decimal reference at slot %r" % first_diff
    # whole final state against the fp64 oracle
    want, wst = oracle.solve(A, b, c, True, kind=oracle.FP64, threads=8, want_trace=False)
    wA, wb, wc, wv, wperm = wst.read()
    out = (C.c_uint64 * 3)()
    assert L.lpx_state_checksum(keep, out) == 0
    assert (int(out[0]), int(out[1]), int(out[2])) == checksum_host(wA, wb, wc)
    assert bits(np.array([res.objective]))[0] == bits(np.array([want["objective"]]))[0]
    assert list(perm) == list(wperm)
    L.lpx_state_destroy(keep)


def test_get_dual_on_device(lps, reference_vectors):                # LPStandardFormSpec.groovy:6-25
    g = reference_vectors["get_dual"]
    names = {i: "x%d" % (i + 1) for i in range(4)}
    form = lps.LPStandardForm(g["A"], g["b"], g["c"], names, {v: k for k, v in names.items()}, 3, 4, g["maximize"])
    dual = form.get_dual()
    assert dual.A.tolist() == g["dualA"]
    assert dual.b.tolist() == g["c"] and dual.c.tolist() == g["b"]
    assert (dual.m, dual.n, dual.maximize) == (g["dual_m"], g["dual_n"], g["dual_maximize"])
    rng = np.random.default_rng(0)
    big = rng.random((300, 1000))
    d2 = lps.LPStandardForm(big, np.ones(300), np.ones(1000)).get_dual()
    assert np.array_equal(d2.A, big.T)


def test_solution_vector_and_names_after_solve(lps, oracle):
    """x* (the reference's commented-out printSolution, LPSolver.java:344-374) is feasible and attains v."""
    A, b, c = dense_lp(60, 90, seed=21)
    solver = lps.LPSolver()
    ans = solver.solve(lps.LPStandardForm(A, b, c, maximize=True))
    x = solver.last.x
    assert np.all(x >= 0) and np.all(A @ x <= b * (1 + 1e-12) + 1e-9)
    assert abs(c @ x - solver.last.objective) <= 1e-9 * max(1, abs(solver.last.objective))
    assert abs(float(ans) - solver.last.objective) <= 5e-7


# ------------------------------------------------------------------------------------ fuzz + full-size properties
def test_fuzz_small_lps_all_outcomes(lps, oracle):
    """250 random small LPs (feasible / infeasible start, unbounded, infeasible, degenerate integer data,
    1 <= m,n <= 24): status, pivot counts, x0 slot, objective bits and basis vs the fp64 oracle."""
    rng = np.random.default_rng(2026)
    seen = {}
    for t in range(250):
        m, n = int(rng.integers(1, 25)), int(rng.integers(1, 25))
        if t % 3 == 0:
            A = rng.integers(-2, 4, size=(m, n)).astype(float)
            b = rng.integers(-2, 6, size=m).astype(float)
            c = rng.integers(-1, 4, size=n).astype(float)
        else:
            A = np.round(rng.uniform(-3, 5, size=(m, n)), 3)
            b = np.round(rng.uniform(-2 if t % 3 == 1 else 0.5, 9, size=m), 3)
            c = np.round(rng.uniform(-2, 4, size=n), 3)
        mx = bool(rng.integers(0, 2))
        want, wst = oracle.solve(A, b, c, mx, kind=oracle.FP64, want_trace=False)
        solver = lps.LPSolver()
        try:
            solver.solve(lps.LPStandardForm(A, b, c, maximize=mx))
        except (lps.LPException, IndexError, ZeroDivisionError):
            pass
        got = solver.last
        key = (want["status"], want["phase1_used"])
        seen[key] = seen.get(key, 0) + 1
        assert got.status == want["status"], (t, m, n)
        assert (got.phase1_used, got.pivots_phase1, got.pivots_phase2, got.x0_slot) == \
            (want["phase1_used"], want["pivots1"], want["pivots2"], want["x0_slot"]), (t, m, n)
        assert bits(np.array([got.objective]))[0] == bits(np.array([want["objective"]]))[0], (t, m, n)
        if got.status == 0:
            assert list(got.perm) == list(wst.read()[4]), (t, m, n)
    assert seen.get((0, False), 0) > 20 and seen.get((0, True), 0) > 5 and seen.get((2, True), 0) > 5, seen


def test_cfg3_size_pivots_match_fp64_oracle(lps, oracle):
    """BASELINE cfg3 size (m=8192, n=16384, 1 GiB tableau, beyond the Infinity Cache): 12 pivots of the device
    loop, then position-keyed checksums of A, b, c plus v and the permutation against the fp64 oracle."""
    import bench
    from linear_programming_solver_amd.lp_state import checksum_host
    m, n = 8192, 16384
    A, b, c = bench.gen_rows(m, n, 1, 0, m)
    st = lps.LPState(A, b, c)
    status, pivots, _ = st.simplex_loop(max_pivots=12)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop(max_pivots=12, threads=16)
    assert (status, pivots) == (want["status"], want["pivots"]) == (9, 12)
    wA, wb, wc, wv, wperm = ref.read()
    assert st.checksum() == checksum_host(wA, wb, wc)
    _, gb, gc, gv, gperm = st.read(want_A=False)
    assert gv == wv and list(gperm) == list(wperm)
    st.close()


def test_cfg4_size_properties_and_sharding_invariance(lps):
    """BASELINE cfg4 size (m=32768, n=16384, 4 GiB): too large for a CPU replay inside a test, so check
    size-independent properties: the objective never decreases and b stays >= 0 along the pivots (feasible
    start, maximisation), perm stays a permutation, and an 8-way row-block sharded run on the same GPU lands
    on the same checksum-of-checksums, b, c, v and permutation as the unsharded run."""
    import torch
    import bench
    from linear_programming_solver_amd.sharded import HipShardEngine, LocalExchange, row_block, sharded_simplex_loop
    m, n = 32768, 16384
    A, b, c = bench.gen_rows(m, n, 1, 0, m)
    st = lps.LPState(A, b, c)
    last_v = 0.0
    for _ in range(3):
        status, pivots, _ = st.simplex_loop(max_pivots=5)
        assert (status, pivots) == (9, 5)
        _, gb, gc, gv, gperm = st.read(want_A=False)
        assert gv >= last_v and np.all(gb >= 0)
        last_v = gv
    assert sorted(gperm.tolist()) == list(range(n + m))
    single = (st.checksum(), gb.copy(), gc.copy(), gv, gperm.copy())
    st.close()
    nsh = 8
    stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
    engines = []
    for r in range(nsh):
        r0, r1 = row_block(m, nsh, r)
        engines.append(HipShardEngine(A[r0:r1], b[r0:r1], c, r0, m, nsh, device=0, stream=stream, comm_stream=comm,
                                      pipeline=2))
    for rep in range(3):
        status, pivots, _ = sharded_simplex_loop(engines, LocalExchange(), max_pivots=5, poll_every=4,
                                                 lookahead=(rep != 1))
        assert (status, pivots) == (9, 5)
    sums = [e.checksum() for e in engines]
    mask = (1 << 64) - 1
    assert sum(s[0] for s in sums) & mask == single[0][0]          # A: checksum of checksums
    assert sum(s[1] for s in sums) & mask == single[0][1]          # b (each shard sums its own rows)
    assert all(s[2] == single[0][2] for s in sums)                  # c is replicated
    sb = np.concatenate([e.read(want_A=False)[1] for e in engines])
    assert np.array_equal(bits(sb), bits(single[1]))
    _, _, sc, sv, sperm = engines[0].read(want_A=False)
    assert np.array_equal(bits(sc), bits(single[2])) and sv == single[3] and list(sperm) == list(single[4])
    for e in engines:
        e.close()


def test_create_from_device_pointers(lps, oracle):
    """lpx_state_create_from_device: inputs already resident in HBM (torch CUDA tensors here)."""
    import ctypes as C
    import torch
    from linear_programming_solver_amd import _lib
    m, n = 40, 70
    A, b, c = dense_lp(m, n, seed=77)
    dA = torch.from_numpy(A).cuda(); db = torch.from_numpy(b).cuda(); dc = torch.from_numpy(c).cuda()
    torch.cuda.synchronize()
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.lpx_state_create_from_device(m, n, C.c_void_p(dA.data_ptr()), n, C.c_void_p(db.data_ptr()),
                                        C.c_void_p(dc.data_ptr()), 0.0, None, 0, m, 0, C.byref(h))
    assert rc == 0, _lib.last_error()
    assert L.lpx_state_set_option(h, _lib.OPTIONS["fused"], 1 if _lib.DEFAULT_FUSED else 0) == 0
    piv, st = C.c_int64(), C.c_int32()
    assert L.lpx_simplex_loop(h, -1, C.byref(piv), C.byref(st), None) == 0
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop()
    assert (st.value, piv.value) == (want["status"], want["pivots"])
    gA = np.zeros((m, n)); gb = np.zeros(m); gc = np.zeros(n); gv = C.c_double(); gp = np.zeros(n + m, dtype=np.int32)
    L.lpx_state_read(h, gA.ctypes.data_as(_lib.dp), n, gb.ctypes.data_as(_lib.dp), gc.ctypes.data_as(_lib.dp),
                     C.byref(gv), gp.ctypes.data_as(_lib.ip))
    assert_state_bits_equal((gA, gb, gc, gv.value, gp), ref.read())
    assert np.array_equal(dA.cpu().numpy(), A)          # the caller's device buffers are never written
    L.lpx_state_destroy(h)


def test_two_handles_from_two_host_threads(lps, oracle):
    """Different handles may be driven from different host threads (include/lpx.h threading contract)."""
    import threading
    results = {}

    def work(k):
        A, b, c = dense_lp(90 + k, 150, seed=500 + k)
        st = lps.LPState(A, b, c)
        status, pivots, _ = st.simplex_loop()
        results[k] = (status, pivots, st.read())
        st.close()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(3):
        A, b, c = dense_lp(90 + k, 150, seed=500 + k)
        ref = oracle.State(A, b, c, kind=oracle.FP64)
        want = ref.simplex_loop()
        assert results[k][:2] == (want["status"], want["pivots"])
        assert_state_bits_equal(results[k][2], ref.read(), "thread %d" % k)


def test_leading_dimension_and_lda_validation(lps):
    import ctypes as C
    from linear_programming_solver_amd import _lib
    L = _lib.lib()
    m, n, lda = 5, 7, 11
    rng = np.random.default_rng(4)
    buf = rng.random((m, lda)); b = np.ones(m) * 3; c = rng.random(n)
    h = C.c_void_p()
    assert L.lpx_state_create(m, n, buf.ctypes.data_as(_lib.dp), lda, b.ctypes.data_as(_lib.dp),
                              c.ctypes.data_as(_lib.dp), 0.0, None, 0, m, 0, C.byref(h)) == 0
    out = np.full((m, lda), -7.0)
    L.lpx_state_read(h, out.ctypes.data_as(_lib.dp), lda, None, None, None, None)
    assert np.array_equal(out[:, :n], buf[:, :n]) and np.all(out[:, n:] == -7.0)   # only n columns per row touched
    L.lpx_state_destroy(h)
    h2 = C.c_void_p()
    assert L.lpx_state_create(m, n, buf.ctypes.data_as(_lib.dp), n - 1, b.ctypes.data_as(_lib.dp),
                              c.ctypes.data_as(_lib.dp), 0.0, None, 0, m, 0, C.byref(h2)) == _lib.BAD_ARGUMENT
    assert L.lpx_state_create(-1, n, None, n, None, c.ctypes.data_as(_lib.dp), 0.0, None, 0, 0, 0,
                              C.byref(h2)) == _lib.BAD_ARGUMENT
    assert L.lpx_state_create(m, n, buf.ctypes.data_as(_lib.dp), lda, b.ctypes.data_as(_lib.dp),
                              c.ctypes.data_as(_lib.dp), 0.0, None, 0, m, 99, C.byref(h2)) == _lib.BAD_ARGUMENT


# ------------------------------------------------------------------------------------ opt-in Dantzig pricing
def test_dantzig_pricing_bit_exact_vs_oracle_with_the_same_rule(lps, oracle):
    """pricing="dantzig" (largest c[j], lowest slot on ties) is an opt-in extension that leaves the reference's
    pivot sequence; its checker is the oracle running the same rule."""
    import torch
    from linear_programming_solver_amd.sharded import HipShardEngine, LocalExchange, row_block, sharded_simplex_loop
    for (m, n) in [(60, 90), (200, 1100), (257, 300)]:
        A, b, c = dense_lp(m, n, seed=31 * m + n)
        st = lps.LPState(A, b, c, pricing="dantzig")
        ref = oracle.State(A, b, c, kind=oracle.FP64, pricing=1)
        assert st.get_entering() == ref.get_entering() == int(np.argmax(c))
        status, pivots, _ = st.simplex_loop()
        want = ref.simplex_loop()
        assert (status, pivots) == (want["status"], want["pivots"])
        assert_state_bits_equal(st.read(), ref.read(), "dantzig %dx%d" % (m, n))
        first = oracle.State(A, b, c, kind=oracle.FP64).simplex_loop()
        assert pivots < first["pivots"]                       # the point of the option
        st.close()
        for lookahead in (False, True):                        # and on row-block shards, both loop forms
            stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
            engines = []
            for r in range(3):
                r0, r1 = row_block(m, 3, r)
                engines.append(HipShardEngine(A[r0:r1], b[r0:r1], c, r0, m, 3, stream=stream, comm_stream=comm,
                                              pricing="dantzig"))
            s2, p2, _ = sharded_simplex_loop(engines, LocalExchange(), poll_every=5, lookahead=lookahead)
            assert (s2, p2) == (want["status"], want["pivots"])
            got = np.vstack([e.read()[0] for e in engines])
            assert np.array_equal(bits(got), bits(ref.read()[0]))
            for e in engines:
                e.close()


def test_dantzig_pricing_through_solve_with_phase1(lps, oracle):
    rng = np.random.default_rng(8)
    for t in range(12):
        m, n = int(rng.integers(3, 30)), int(rng.integers(3, 30))
        A = np.round(rng.uniform(-3, 5, size=(m, n)), 3)
        b = np.round(rng.uniform(-2, 9, size=m), 3)
        c = np.round(rng.uniform(-2, 4, size=n), 3)
        want, wst = oracle.solve(A, b, c, True, kind=oracle.FP64, want_trace=False, pricing=1)
        solver = lps.LPSolver(pricing="dantzig")
        try:
            solver.solve(lps.LPStandardForm(A, b, c, maximize=True))
        except (lps.LPException, IndexError):
            pass
        got = solver.last
        assert (got.status, got.pivots_phase1, got.pivots_phase2, got.x0_slot) == \
            (want["status"], want["pivots1"], want["pivots2"], want["x0_slot"]), t
        assert bits(np.array([got.objective]))[0] == bits(np.array([want["objective"]]))[0], t


def test_reference_slack_and_aux_conversions(lps, reference_vectors):   # LPSolverSpec.groovy:37-74
    g = reference_vectors["aux_lp_conversion"]
    names = {i: "x%d" % (i + 1) for i in range(5)}
    form = lps.LPStandardForm(g["A"], g["b"], g["c"], names, {v: k for k, v in names.items()}, 5, 5, True)
    st = lps.LPSolver().convert_into_aux_lp(form)
    A, b, c, v, perm = st.read()
    assert A.tolist() == g["resA"] and c.tolist() == g["resC"] and b.tolist() == g["b"]
    assert len(st.coefficients) == len(st.variables) == 11
    assert "x0" in st.coefficients and st.variables[5] == "x0"
    st.close()
    form2 = lps.LPStandardForm(np.ones((4, 4)), np.ones(4), np.ones(4),
                               {0: "x0", 1: "x1", 2: "x4", 3: "x6"}, {"x0": 0, "x1": 1, "x4": 2, "x6": 3}, 4, 4, True)
    st2 = lps.LPSolver().convert_into_slack_form(form2)
    assert len(st2.variables) == 8 and len(st2.coefficients) == 8
    assert [st2.variables[4 + i] for i in range(4)] == ["x2", "x3", "x5", "x7"]
    # solveAuxLP on the Spec's 4x3 aux state leaves v == 0 (LPSolverSpec.groovy:113-124)
    h = reference_vectors["solve_aux_lp"]
    aux = lps.LPState(h["A"], h["b"], h["c"])
    aux.pivot(h["index_of_x0"], h["min_in_b"])
    status, _, x0 = aux.simplex_loop(track_slot=h["min_in_b"] + 3)
    assert status == 0 and aux.v == h["resV"] and x0 == 1          # x0 ends in slot 1 (logs/lp_solver.log:196)
    aux.close(); st2.close()


def test_reference_restore_initial_lp_vector(lps, reference_vectors):     # LPSolverSpec.groovy:126-149
    g = reference_vectors["restore_initial_lp"]
    ident = {"x1": 0, "x2": 1, "x3": 2, "x4": 3, "x5": 4, "x6": 5, "x0": 6}   # originals, slacks, x0 = n+m
    perm = [ident[s] for s in g["aux_names"]]
    names = {s: nm for s, nm in enumerate(g["aux_names"])}
    aux = lps.LPState(g["auxA"], g["auxB"], g["auxC"], 0.0, names, {v: k for k, v in names.items()}, 4, 3, perm=perm)
    initial = lps.LPStandardForm(np.zeros((4, 2)), np.zeros(4), g["initial_c"], {0: "x1", 1: "x2"},
                                 {"x1": 0, "x2": 1}, 4, 2, True)
    res = lps.LPSolver().restore_initial_lp(aux, initial, g["index_of_x0"])
    A, b, c, v, p = res.read()
    assert A.tolist() == g["resA"] and b.tolist() == g["resB"] and c.tolist() == g["resC"] and v == g["resV"]
    assert res.variables == {s: nm for s, nm in enumerate(g["res_names"])}
    assert res.coefficients == {nm: s for s, nm in enumerate(g["res_names"])}
    res.close()


def test_restore_index_fault_is_reported(lps, decimal_goldens):
    """A golden that triggers the reference's ArrayIndexOutOfBoundsException in restoreInitialLP."""
    case = next(c for c in decimal_goldens["lp_cases"] if c["status"] == 6)
    m, n = case["m"], case["n"]
    A = np.array([float(x) for x in case["A"]]).reshape(m, n)
    solver = lps.LPSolver()
    with pytest.raises(IndexError):
        solver.solve(lps.LPStandardForm(A, [float(x) for x in case["b"]], [float(x) for x in case["c"]],
                                        maximize=case["maximize"]))
    assert solver.last.status == 6 and solver.last.x0_slot == case["x0_slot"]


# ------------------------------------------------------------------------------------ blocked pivoting
@pytest.mark.parametrize("block", [2, 3, 4, 8, 16, 21, 32, 33, 48, 64])
@pytest.mark.parametrize("shape", [(50, 80), (257, 300), (200, 1100), (9, 2100)])
def test_blocked_pivoting_is_bit_identical(lps, oracle, shape, block):
    """K pivot decisions from the stale tableau + one K-fold sweep must equal K separate updates bit for bit:
    budgets that are not multiples of K, resumed loops, and runs to optimality."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=13 * m + n)
    st = lps.LPState(A, b, c, block=block)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (1, block, block + 1, 3 * block - 1, 0, -1):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, block, budget)
        assert_state_bits_equal(st.read(), ref.read(), "block %d budget %d of %s" % (block, budget, shape))
    st.close()


@pytest.mark.parametrize("knobs", [
    {"overlap": 0},                              # k_block_chain, then the in-place sweep, one after the other
    {"chain": 0},                                # three launches per decision (the form the shards use)
    {"overlap_serial": 1},                       # out-of-place sweeps without concurrency
    {"overlap_mask": 0},                         # decisions beside the sweep without CU masks
    {"chain_wgs": 1}, {"chain_wgs": 3},          # several rows / columns per thread in k_block_chain
    {"chain_wgs": 7, "overlap": 0},
    {"chain_fences": 3},                         # release + acquire at every grid barrier (the conservative form)
    {"sweep_rows": 64}, {"sweep_rows": 192}, {"sweep_rows": 4096},   # one chunk / three chunks / one run per strip
], ids=lambda k: ",".join("%s=%s" % kv for kv in sorted(k.items())))
def test_blocked_loop_forms_are_bit_identical(lps, oracle, knobs):
    """Every form of the blocked loop (default: decisions one block ahead of out-of-place sweeps) gives the
    one-pass-per-pivot result bit for bit; the forms are selected through the handle (lpx_state_set_option)."""
    for (m, n), block in (((300, 700), 32), ((1100, 260), 16), ((64, 2100), 8), ((400, 1300), 64)):
        A, b, c = dense_lp(m, n, seed=7 * m + n)
        st = lps.LPState(A, b, c, block=block, options=knobs)
        for k, v in knobs.items():
            assert st.get_option(k) == v
        ref = oracle.State(A, b, c, kind=oracle.FP64)
        for budget in (block, 2 * block + 3, 5 * block, 1, -1):
            status, pivots, _ = st.simplex_loop(max_pivots=budget)
            want = ref.simplex_loop(max_pivots=budget)
            assert (status, pivots) == (want["status"], want["pivots"]), (knobs, m, n, block, budget)
            assert_state_bits_equal(st.read(), ref.read(), "%s block %d budget %d" % (knobs, block, budget))
        st.close()


def test_option_validation(lps):
    st = lps.LPState([[1.0, 2.0]], [1.0], [1.0, 1.0])
    with pytest.raises(ValueError):
        st.set_option("chain_fences", 7)
    with pytest.raises(ValueError):
        st.set_option(99, 0)
    with pytest.raises(ValueError):
        st.set_option("update_u", 3)
    st.set_option("block", 4)
    assert st.block() == 4 and st.get_option("block") == 4
    st.close()
    with pytest.raises(ValueError):
        lps.LPState(np.ones((3, 2)), [1.0, 1.0], [1.0, 1.0])      # wrong-shaped A is an error, not a zero tableau


def _timed_form_vs_oracle(lps, oracle, m, n, budgets, options=None, threads=16):
    """The loop form bench.py times (default options: blocks of 32 decisions in the persistent decision kernel, one
    block ahead of out-of-place non-temporal sweeps) against the fp64 oracle at full size: status, pivot count, v,
    perm, b, c bit for bit and the position-keyed checksum of A (host side in row chunks)."""
    import bench
    from linear_programming_solver_amd.lp_state import checksum_host
    A, b, c = bench.gen_rows(m, n, 1, 0, m)
    st = lps.LPState(A, b, c, options=options)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    del A
    for budget in budgets:
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=threads)
        assert (status, pivots) == (want["status"], want["pivots"]) == (9, budget)
        wA, wb, wc, wv, wperm = ref.read()
        _, gb, gc, gv, gperm = st.read(want_A=False)
        assert gv == wv and list(gperm) == list(wperm), budget
        assert np.array_equal(bits(gb), bits(wb)) and np.array_equal(bits(gc), bits(wc)), budget
        mask = (1 << 64) - 1
        sa = 0
        for r0 in range(0, m, 2048):
            sa = (sa + checksum_host(wA[r0:r0 + 2048], wb[:0], wc[:0], row0=r0)[0]) & mask
        del wA
        assert st.checksum()[0] == sa, "tableau differs from the oracle after a budget of %d" % budget
    info = st.info()
    st.close()
    ref.close()
    return info


def test_cfg3_timed_form_200_pivots_vs_fp64_oracle(lps, oracle, arith):
    """BASELINE cfg3 (8192 x 16384): 70 + 200 pivots through the default loop — full blocks (32 pivots; 64 on the matrix
    cores in the fused arithmetic, by size from ~0.85 GiB since round 5), the wide decision kernel beside the sweeps, a
    budget tail folded into the last block."""
    info = _timed_form_vs_oracle(lps, oracle, 8192, 16384, (70, 200))
    assert info["block"] == (64 if arith == "fused" else 32) and info["overlapped"] == 1
    assert info["chain_wgs"] <= info["chain_resident_max"]
    if info["chain_stream_masked"]:   # by size: 8 CUs per XCD for the decisions, one column per thread (workgroup 0 + 63)
        assert info["chain_resident_max"] == 64 and info["chain_wgs"] == 64, info
    print("cfg3 placement:", info)


@pytest.mark.parametrize("cus,resident", [(4, 32), (6, 32), (8, 64), (12, 96)])
def test_decision_cus_option_vs_fp64_oracle(lps, oracle, cus, resident):
    """Option chain_cus: CUs per XCD of the decision stream (multiples of 4 only — 6 is taken as 4: an uneven share
    of an XCD's shader engines left the grid partly non-resident), one row / column per thread within them, on a
    4096 x 12288 tableau (48 + 1 workgroups wanted) through two full blocks and a tail against the fp64 oracle."""
    info = _timed_form_vs_oracle(lps, oracle, 4096, 12288, (45,), options={"chain_cus": cus, "block": 16})
    assert info["overlapped"] == 1 and info["chain_wgs"] <= info["chain_resident_max"]
    if info["chain_stream_masked"]:
        # (49 = one column per thread + workgroup 0 on the window in k_block_chain_t's grid; k_block_chain2 needs 48)
        assert info["chain_resident_max"] == resident and info["chain_wgs"] in (min(48, resident), min(49, resident)), info


@pytest.mark.parametrize("side", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape,block", [((4096, 12288), 32), ((2048, 4096), 64), ((1000, 1500), 64)])
def test_fixup_beside_or_behind_the_sweep_vs_oracle(lps, oracle, shape, block, side):
    """Option fixup_side (LPX_OPT_FIXUP_SIDE): the entering columns and pivot rows of a block recomputed from the ring
    (LPState.java:139-164) behind the block's sweep (0) or beside it into images that only a copy kernel takes into the
    tableau afterwards (1: on the sweep's CUs, 2: the decisions' CUs, 3: no mask; 4, the default: by size — 2 from 2 GiB, the
    cfg4 tests, 0 below).  Three full blocks and
    tails (a second call continues on the swapped buffers) against the oracle of the arithmetic mode, bit for bit; the
    1000 x 1500 case has partial strips and rows that are not a multiple of 4 (generic sweep kernels)."""
    m, n = shape
    info = _timed_form_vs_oracle(lps, oracle, m, n, (3 * block + 7, block + 3), options={"fixup_side": side, "block": block})
    assert info["block"] == block and info["overlapped"] == 1


def test_cfg3_one_pass_form_30_pivots_vs_fp64_oracle(lps, oracle):
    """The roofline kernel of north_star itself at HBM size: one pass per pivot (k_select_pivot + k_update<1, nt>, the
    form `bench.py --option block=1` times at 6.3 TB/s) for 30 pivots at cfg3 against the fp64 oracle."""
    info = _timed_form_vs_oracle(lps, oracle, 8192, 16384, (30,), options={"block": 1})
    assert info["block"] == 1 and info["nontemporal"] == 1


@pytest.mark.parametrize("block", [1, 2])
def test_cfg4_onepass_forms_12_pivots_vs_fp64_oracle(lps, oracle, block):
    """The kernels behind bench.py's `onepass` object at the HEIGHT it times them at (32768 rows x 16384 columns, 4 GiB:
    k_update<1, nt> = one pass per pivot, the reference's own schedule, LPState.java:150-181; k_update_tiles<2> = two
    pivots per pass) against the oracle of the arithmetic mode, bit for bit."""
    info = _timed_form_vs_oracle(lps, oracle, 32768, 16384, (12,), options={"block": block})
    assert info["block"] == block and info["nontemporal"] == 1


def test_cfg3_sweep_of_blocks_of_32_vs_fp64_oracle(lps, oracle):
    """The steady-state sweep kernel of blocks of 17..32 pivots (k_sweep32_pull: LDS-DMA staging, batches pulled in address
    order) through two full blocks and a tail at cfg3 (a short one, which the tile kernel takes, then one of 28 pivots, which
    it takes padded with identity steps), against the fp64 oracle.  (The kernels it replaced — sweep_form 1 / 2 — are checked
    the same way in tests/test_gpu_variants.py, on the variants library.)"""
    info = _timed_form_vs_oracle(lps, oracle, 8192, 16384, (75, 60), options={"sweep_form": 0, "block": 32})
    assert info["block"] == 32 and info["sweep_kernel_name"] == "k_sweep32_pull"


def test_cfg4_timed_form_70_pivots_vs_fp64_oracle(lps, oracle, arith):
    """BASELINE cfg4 (32768 x 16384, 4 GiB): 25 pivots (the driver's bench command is 5 + 20: a budget that fits one
    block goes through the serial form on the whole chip), then two full K = 32 blocks + a tail, then a full block + a
    partly filled one through the default overlapped loop, every time against the fp64 oracle."""
    info = _timed_form_vs_oracle(lps, oracle, 32768, 16384, (25, 70, 50))   # (the last sweep: 18 pivots, padded to 32)
    # by size: blocks of 32 (k_sweep32_pull); in the fused-arithmetic mode blocks of 64 on the matrix cores from ~2.5 GiB
    # (there the last sweep of this test, 18 pivots of a block of 64, is two generic passes)
    # (the last budget, 50 pivots, fits ONE block of 64: the serial form on the whole chip)
    assert info["block"] == (64 if arith == "fused" else 32) and info["nontemporal"] == 1
    assert info["overlapped"] == (0 if arith == "fused" else 1)
    if arith != "fused":
        assert info["sweep_kernel_name"] == "k_sweep32_pull"
    assert info["chain_wgs"] <= info["chain_resident_max"]
    print("cfg4 placement:", info)


def test_tableau_beyond_4_gib_vs_fp64_oracle(lps, oracle, arith):
    """36864 x 18432 = 5.4 GB per tableau buffer: byte offsets beyond 2^32 in the sweeps (32-bit offsets are per run of
    rows only), the decision kernel's strided column reads and the fix-up, 40 pivots through the default loop."""
    info = _timed_form_vs_oracle(lps, oracle, 36864, 18432, (40,))
    assert info["block"] == (64 if arith == "fused" else 32) and info["overlapped"] == (0 if arith == "fused" else 1)


def test_8_gib_tableau_takes_blocks_of_64_by_size(lps, oracle):
    """40960 x 24576 = 8.05 GB per tableau buffer: above ~7 GiB the by-size choice is blocks of 64 (two-stage sweep,
    64-slot decisions); one full block + a tail of 6 against the fp64 oracle."""
    info = _timed_form_vs_oracle(lps, oracle, 40960, 24576, (70,))
    assert info["block"] == 64 and info["overlapped"] == 1


def test_cfg4_blocks_of_64_vs_fp64_oracle(lps, oracle):
    """Opt-in blocks of 64 at BASELINE cfg4 size: one full block through the two-stage sweep kernel and the 64-slot
    decision kernel, then a tail of 36 (two generic passes), against the fp64 oracle."""
    info = _timed_form_vs_oracle(lps, oracle, 32768, 16384, (100,), options={"block": 64})
    assert info["block"] == 64 and info["overlapped"] == 1


@pytest.mark.parametrize("shape", [(2048, 4096), (4096, 8192), (8192, 2048)])
@pytest.mark.parametrize("fences,block", [(2, 32), (3, 32), (2, 64)])
def test_wide_decision_kernel_vs_oracle(lps, oracle, shape, fences, block):
    """The decision kernel at full width (33 workgroups requested, clamped to what is resident) on 2-8 k-row
    shapes, block 32 (and the 64-slot form), beside the sweeps: workgroup 0's one-way hand-off and the grid barrier
    run wide against the oracle, in both barrier forms (acquire-only default, release + acquire)."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=3 * m + n)
    st = lps.LPState(A, b, c, block=block, options={"chain_wgs": 33, "chain_fences": fences})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (96, 45, 160):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=16)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, fences, budget)
        assert_state_bits_equal(st.read(), ref.read(), "%s fences %d budget %d" % (shape, fences, budget))
    info = st.info()
    assert 8 <= info["chain_wgs"] <= min(33, info["chain_resident_max"]), info
    st.close()


@pytest.mark.parametrize("shape,block,wgs", [((1024, 2048), 1, None), ((1280, 2048), 16, 9), ((2048, 4096), 32, 17),
                                             ((4096, 8192), 32, 32), ((6144, 8192), 32, 32), ((8192, 8192), 32, 32)])
def test_block_and_decision_grid_by_size(lps, oracle, shape, block, wgs):
    """The by-size choices of the default loop (round 5: profiles/r05_block_policy_*.txt; r03_decision_grid.txt): one pass per
    pivot up to ~18 MiB on a handle without a ring, blocks of 16 up to ~28 MiB, 32 above; the decision kernel with one row /
    column per thread (+ the workgroup of the hand-off window), within the 32 CUs of its masked stream.  150 pivots against
    the fp64 oracle."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=m + 7 * n)
    st = lps.LPState(A, b, c)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    assert st.block() == block
    status, pivots, _ = st.simplex_loop(max_pivots=150)
    want = ref.simplex_loop(max_pivots=150, threads=16)
    assert (status, pivots) == (want["status"], want["pivots"])
    assert_state_bits_equal(st.read(), ref.read(), "%s by-size loop" % (shape,))
    info = st.info()
    if wgs is not None and info["chain_stream_masked"]:
        # (k_block_chain2 sizes its own grid: one row / one column per thread, workgroup 0 on the window only — never
        # more than the by-size bound the engine computed)
        form2 = st.get_option("chain_form") == 1
        assert (info["chain_wgs"] <= wgs if form2 else info["chain_wgs"] == wgs) and info["chain_wgs"] <= info["chain_resident_max"], info
    st.close()


def test_decision_kernel_residency_is_bounded(lps):
    """lpx_state_get_info: the decision kernel's grid never exceeds what the CUs of its stream hold at once (its
    workgroups spin at grid barriers), whatever is requested; the placement census names the XCDs it ran on."""
    A, b, c = dense_lp(1024, 2048, seed=5)
    for masked in (1, 0):
        st = lps.LPState(A, b, c, block=16, options={"chain_wgs": 256, "overlap_mask": masked, "chain_trace": 1})
        status, pivots, _ = st.simplex_loop(max_pivots=40)
        assert (status, pivots) == (9, 40)
        info = st.info()
        assert info["chain_wgs_requested"] == 256
        cap = min(256, info["chain_resident_max"])
        assert (1 <= info["chain_wgs"] <= cap if st.get_option("chain_form") == 1 else info["chain_wgs"] == cap)
        assert info["chain_blocks_per_cu"] >= 1
        assert info["chain_xcd_mask"] != 0
        if info["chain_stream_masked"]:
            assert info["chain_resident_max"] == info["chain_blocks_per_cu"] * 32
        tr = st.chain_trace()
        assert tr.shape[1] == 5 and tr.shape[0] >= 1 and np.all(np.diff(tr[0]) >= 0)
        st.close()


def test_solve_infeasible_with_exactly_sized_perm_out(lps):
    """lpx_solve on an LP that ends inside phase 1 must not write past perm_out's n+m entries (the auxiliary LP's
    permutation has n+m+1)."""
    import ctypes as C
    from linear_programming_solver_amd import _lib
    L = _lib.lib()
    A = np.array([[1.0, 1.0], [-1.0, -1.0]])
    b = np.array([1.0, -3.0])         # x1 + x2 <= 1 and x1 + x2 >= 3: infeasible
    c = np.array([1.0, 1.0])
    m, n = 2, 2
    guard = np.full(n + m + 8, 0x5A5A5A5A, dtype=np.int32)
    opts = _lib.SolveOptions()
    opts.fused = 1 if _lib.DEFAULT_FUSED else 0     # the raw C call: the arithmetic mode this test runs in
    opts.max_pivots = -1
    opts.perm_out = guard.ctypes.data_as(_lib.ip)
    res = _lib.SolveResult()
    rc = L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp), n, b.ctypes.data_as(_lib.dp), c.ctypes.data_as(_lib.dp), 1,
                     C.byref(opts), C.byref(res))
    assert rc == _lib.INFEASIBLE and res.phase1_used == 1
    assert np.all(guard[n + m:] == 0x5A5A5A5A), guard
    # and a solve that reaches phase 2 fills exactly n+m entries
    b2 = np.array([4.0, -1.0])
    guard[:] = 0x5A5A5A5A
    rc = L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp), n, b2.ctypes.data_as(_lib.dp), c.ctypes.data_as(_lib.dp), 1,
                     C.byref(opts), C.byref(res))
    assert rc == _lib.OPTIMAL and res.phase1_used == 1
    assert sorted(guard[: n + m].tolist()) == list(range(n + m)) and np.all(guard[n + m:] == 0x5A5A5A5A)


def test_solve_named_dual_with_more_variables_than_names(lps, oracle):
    """solve(getDual()) of a named form with m > n: the reference's getDual names only n of the dual's m variables
    (LPStandardForm.java:139-142); restoreInitialLP then visits only the named ones."""
    A = np.array([[1.0, 2.0], [3.0, 1.0], [1.0, 1.0]])          # m = 3 > n = 2
    form = lps.LPStandardForm(A, [4.0, 5.0, 3.0], [1.0, 1.0], {0: "x1", 1: "x2"}, {"x1": 0, "x2": 1}, maximize=True)
    dual = form.get_dual()
    assert dual.n == 3 and len(dual.coefficients) == 2
    solver = lps.LPSolver()
    ans = solver.solve(dual)          # min 4 y1 + 5 y2 + 3 y3, A^T y <= c -> needs phase 1? (b = c >= 0: no)
    res, _ = oracle.solve(dual.A, dual.b, dual.c, dual.maximize, kind=oracle.FP64, want_trace=False)
    assert str(ans) == res["objective_text"]
    # with a negative right-hand side phase 1 runs and the keySet() order has 2 entries for 3 variables
    dual2 = lps.LPStandardForm(dual.A, [1.0, -1.0], dual.c, dual.variables, dual.coefficients, maximize=True)
    try:
        solver.solve(dual2)
    except Exception as ex:           # whatever the outcome, it must be a solver outcome, not a KeyError
        assert not isinstance(ex, KeyError), ex


def test_blocked_loop_many_blocks_and_dantzig(lps, oracle):
    """Long runs through the overlapped loop (many blocks in flight, both ring halves and both tableau buffers in
    use) for both pricing rules, resumed in pieces whose lengths are not multiples of the block."""
    m, n = 600, 900
    A, b, c = dense_lp(m, n, seed=99)
    for pricing, rule in ((None, 0), ("dantzig", 1)):
        st = lps.LPState(A, b, c, block=32, **({"pricing": pricing} if pricing else {}))
        ref = oracle.State(A, b, c, kind=oracle.FP64, pricing=rule)
        for budget in (500, 333, 64, -1):
            status, pivots, _ = st.simplex_loop(max_pivots=budget)
            want = ref.simplex_loop(max_pivots=budget)
            assert (status, pivots) == (want["status"], want["pivots"]), (pricing, budget)
            assert_state_bits_equal(st.read(), ref.read(), "%s budget %d" % (pricing, budget))
        st.close()


@pytest.mark.parametrize("shape", [(600, 900), (1536, 700)])
def test_blocks_of_64_with_long_ladders(lps, oracle, shape):
    """Dantzig pricing seldom returns to a slot, so hardly any decision restarts inside the pending pivots: with blocks of
    64 in the overlapped loop a decision's ladder then runs through up to 128 pending pivots — more than the one window of
    eight live chunks a thread of k_block_chain2_t<64, ...> holds in registers: the second window (a second round trip) and
    the boundary value between the two blocks' pivots in either window.  Both rules, budgets that are no multiples of 64."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=m + 7 * n)
    for pricing, rule in (("dantzig", 1), (None, 0)):
        st = lps.LPState(A, b, c, block=64, **({"pricing": pricing} if pricing else {}))
        ref = oracle.State(A, b, c, kind=oracle.FP64, pricing=rule)
        for budget in (200, 333, 129):
            status, pivots, _ = st.simplex_loop(max_pivots=budget)
            want = ref.simplex_loop(max_pivots=budget)
            assert (status, pivots) == (want["status"], want["pivots"]), (pricing, budget)
            assert_state_bits_equal(st.read(), ref.read(), "%s budget %d of %s" % (pricing, budget, shape))
            if status != 9:
                break
        assert st.info()["block"] == 64
        st.close()


def test_arithmetic_by_size_and_the_switch():
    """LPX_OPT_FUSED = 2 (the library's default): fused multiply-add updates on an unsharded tableau of 0.5 GiB and more, the
    reference's two roundings below; 0 / 1 force a mode; lpx_state_info.arith_fused reports it, and the bits follow: 40 pivots
    of a 4096 x 16384 LP (exactly 0.5 GiB) equal the fused oracle by default and the plain one with the opt-out."""
    import linear_programming_solver_amd as pkg
    from oracle import pyoracle as orc
    prev = pkg.set_default_arithmetic("auto")
    try:
        A, b, c = dense_lp(64, 128, seed=3)
        st = pkg.LPState(A, b, c)
        assert st.get_option("fused") == 2 and st.info()["arith_fused"] == 0
        st.set_option("fused", 1)
        assert st.info()["arith_fused"] == 1
        st.close()
        m, n = 4096, 16384
        A, b, c = dense_lp(m, n, seed=8)
        for opts, kind, want in (({}, orc.FP64_FUSED, 1), ({"fused": 0}, orc.FP64, 0)):
            st = pkg.LPState(A, b, c, options=opts)
            assert st.info()["arith_fused"] == want, opts
            ref = orc.State(A, b, c, kind=kind)
            status, pivots, _ = st.simplex_loop(max_pivots=40)
            wantr = ref.simplex_loop(max_pivots=40, threads=8)
            assert (status, pivots) == (wantr["status"], wantr["pivots"])
            _, gb, gc, gv, gperm = st.read(want_A=False)
            _, wb, wc, wv, wperm = ref.read()
            assert gv == wv and list(gperm) == list(wperm) and np.array_equal(bits(gb), bits(wb)) and np.array_equal(bits(gc), bits(wc)), opts
            st.close()
            ref.close()
    finally:
        pkg.set_default_arithmetic(prev)


def test_blocked_pivoting_degenerate_unbounded_and_tracking(lps, oracle):
    # ties everywhere
    m, n = 70, 40
    A = np.ones((m, n)); b = np.full(m, 3.0); c = np.arange(n, 0, -1).astype(float)
    st = lps.LPState(A, b, c, block=4)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    status, pivots, _ = st.simplex_loop()
    want = ref.simplex_loop()
    assert (status, pivots) == (want["status"], want["pivots"])
    assert_state_bits_equal(st.read(), ref.read())
    st.close()
    # unbounded in the middle of a block
    st = lps.LPState([[1.0, 0.0]], [1.0], [1.0, 1.0], block=8)
    assert st.simplex_loop()[:2] == (1, 1)
    st.close()
    # x0 tracking through a blocked phase-1 loop (LPSolverSpec.groovy:113-124 / logs/lp_solver.log:196)
    A = [[1, 0, -1], [-1, 0, -1], [0, 1, -1], [0, -1, -1]]
    aux = lps.LPState(A, [10, -2, 10, -2], [0, 0, -1], block=2)
    aux.pivot(2, 1)
    status, _, x0 = aux.simplex_loop(track_slot=1 + 3)
    assert status == 0 and aux.v == 0 and x0 == 1
    aux.close()


def test_blocked_pivoting_with_dantzig_and_full_solve(lps, oracle):
    A, b, c = dense_lp(300, 700, seed=99)
    st = lps.LPState(A, b, c, pricing="dantzig", block=8)
    ref = oracle.State(A, b, c, kind=oracle.FP64, pricing=1)
    status, pivots, _ = st.simplex_loop()
    want = ref.simplex_loop()
    assert (status, pivots) == (want["status"], want["pivots"])
    assert_state_bits_equal(st.read(), ref.read())
    st.close()


@pytest.mark.parametrize("block", [2, 5, 16, 32])
@pytest.mark.parametrize("nshards,shape,budget", [(2, (64, 100), -1), (4, (130, 513), 25), (8, (257, 2100), 12),
                                                  (3, (10, 40), -1), (5, (1000, 260), 40)])
def test_blocked_pivoting_on_shards_matches_oracle(lps, oracle, nshards, shape, budget, block):
    """Blocked pivoting over row-block shards (all shards on this GPU, LocalExchange): every decision's candidate
    comes from the stale shard + corrections, the winner's stale row is saved on its owner for the fix-up."""
    import torch
    from linear_programming_solver_amd.sharded import HipShardEngine, LocalExchange, row_block, sharded_simplex_loop
    m, n = shape
    A, b, c = dense_lp(m, n, seed=m + n)
    stream = torch.cuda.Stream()
    engines = []
    for r in range(nshards):
        r0, r1 = row_block(m, nshards, r)
        engines.append(HipShardEngine(A[r0:r1], b[r0:r1], c, r0, m, nshards, device=0, stream=stream,
                                      comm_stream=stream, pipeline=1))
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for _ in range(2):
        status, pivots, _ = sharded_simplex_loop(engines, LocalExchange(), max_pivots=budget, poll_every=32, block=block)
        want = ref.simplex_loop(max_pivots=budget)
        assert (status, pivots) == (want["status"], want["pivots"])
    wA, wb, wc, wv, wperm = ref.read()
    for e in engines:
        gA, gb, gc, gv, gperm = e.read()
        r0 = e.row0
        assert np.array_equal(bits(gA), bits(wA[r0:r0 + e.m_local]))
        assert np.array_equal(bits(gb), bits(wb[r0:r0 + e.m_local]))
        assert np.array_equal(bits(gc), bits(wc)) and gv == wv and list(gperm) == list(wperm)
        e.close()


@pytest.mark.skipif(not os.environ.get("LPX_SOAK"), reason="opt-in soak: LPX_SOAK=<repetitions> (150 takes about a minute)")
def test_soak_repeated_runs_on_the_shape_that_exposed_the_stale_read(lps, oracle):
    """scripts/flake_hunt.py as a test: the 8192 x 2048 shape on which round 2's hand-issued register loads returned a
    stale batch once in ~15 000 workgroup runs (11 of 150 repetitions).  One LP, fresh handles, the default loop in two
    budget pieces (a full block + tails), every final tableau compared entry by entry with the fp64 oracle's."""
    reps = int(os.environ["LPX_SOAK"])
    m, n = 8192, 2048
    A, b, c = dense_lp(m, n, seed=3 * m + n)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (96, 45):
        ref.simplex_loop(max_pivots=budget, threads=16)
    want = ref.read()
    ref.close()
    bad = []
    for rep in range(reps):
        st = lps.LPState(A, b, c)
        for budget in (96, 45):
            st.simplex_loop(max_pivots=budget)
        got = st.read()
        st.close()
        if not (np.array_equal(bits(got[0]), bits(want[0])) and got[3] == want[3] and list(got[4]) == list(want[4])):
            bad.append(rep)
    assert not bad, "%d of %d repetitions differ from the oracle: %s" % (len(bad), reps, bad[:20])


@pytest.mark.parametrize("shape", [(1000, 2100), (4100, 1024), (8, 512), (2052, 4100)])
@pytest.mark.parametrize("block,form,kernel", [(21, 0, "k_sweep32_pull"), (32, 0, "k_sweep32_pull"),
                                               (40, 0, "k_sweep64_one"), (64, 0, "k_sweep64_one")])
def test_pulled_sweep_kernels_on_ragged_shapes(lps, oracle, shape, block, form, kernel):
    """The ticket-pulling sweep kernels (one wave per 128-column sub-strip for blocks up to 32; one wave per 64-column
    sub-strip for blocks of 33..64 — these heights are no multiples of 16, so also in the fused mode) on shapes with a partial last
    strip, a last batch count that is not a multiple of anything, fewer batches than workers, and blocks that are only
    partly filled (21 of 32, 40 of 64 steps are real, the rest identities): bit-exact vs the fp64 oracle after every
    budget; the engine must report the kernel that is expected to have run."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=11 * m + n)
    st = lps.LPState(A, b, c, block=block, options={"sweep_form": form})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (block - 1, 2 * block + 5, block - 1):   # a full block of `block` decisions (block - 1 pivots + the probe)
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, block, budget)
        assert_state_bits_equal(st.read(), ref.read(), "block %d budget %d of %s" % (block, budget, shape))
    if status == 9:   # still running: the last sweep applied block - 1 >= 17 pivots
        assert st.info()["sweep_kernel_name"] == kernel, st.info()
    st.close()


@pytest.mark.parametrize("overlap,nt", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_blocks_of_64_in_place_and_out_of_place_with_and_without_nt(lps, oracle, arith, overlap, nt):
    """The four instantiations of the sweep of a block of 64 (in place in the serial loop, out of place beside the
    decisions; with and without the non-temporal hints) on a tableau with more tickets than workers, a partial last
    strip and a partly filled last block: in the fused mode k_sweep64_mfma2, whose tickets past the end re-read the
    last tile and have their stores dropped by the buffer range check — in place that tile may be rewritten by its
    owner meanwhile."""
    m, n = 4096, 5000
    A, b, c = dense_lp(m, n, seed=4242)
    st = lps.LPState(A, b, c, block=64, options={"overlap": overlap, "nt": nt})
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (64, 150, 40):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), (overlap, nt, budget)
        assert_state_bits_equal(st.read(), ref.read(), "overlap %d nt %d budget %d" % (overlap, nt, budget))
    info = st.info()
    assert info["nontemporal"] == nt and info["block"] == 64, info
    if arith == "fused":
        assert info["sweep_kernel_name"] == "k_sweep64_mfma2", info
    st.close()


@pytest.mark.parametrize("shape", [(1024, 2112), (2048, 4100), (4096, 1024), (16, 512)])
@pytest.mark.parametrize("form", [0, 3])
@pytest.mark.parametrize("block", [40, 64])
def test_blocks_of_33_to_64_with_16_row_tiles(lps, oracle, arith, shape, block, form):
    """Tableaus whose height is a multiple of 16: in the fused-arithmetic mode blocks of 33..64 go through the matrix
    cores (k_sweep64_mfma: v_mfma_f64_16x16x4 is a chain of fused multiply-adds in pivot order, so the bits are those of
    64 v_fma_f64 steps), in the default arithmetic through k_sweep64_one; partly filled blocks, a partial last strip,
    fewer tiles than workers: bit-exact against the oracle of the mode after every budget."""
    m, n = shape
    A, b, c = dense_lp(m, n, seed=17 * m + n)
    st = lps.LPState(A, b, c, block=block, options={"sweep_form": form})   # (0: by mode — the matrix cores when fused; 3: k_sweep64_one)
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    for budget in (block - 1, 2 * block + 5, block - 1):
        status, pivots, _ = st.simplex_loop(max_pivots=budget)
        want = ref.simplex_loop(max_pivots=budget, threads=8)
        assert (status, pivots) == (want["status"], want["pivots"]), (shape, block, budget)
        assert_state_bits_equal(st.read(), ref.read(), "block %d budget %d of %s" % (block, budget, shape))
    if status == 9 and n >= 512:
        want_kernel = "k_sweep64_mfma2" if (arith == "fused" and form == 0) else "k_sweep64_one"   # (form 3: the vector kernel in both modes)
        assert st.info()["sweep_kernel_name"] == want_kernel, st.info()
    st.close()
