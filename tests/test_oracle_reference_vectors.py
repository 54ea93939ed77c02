"""Pins the oracle (oracle/lp_oracle.hpp: the decimal-15, the fp64 and the fused-fp64 instantiation) against every
known-answer vector the reference's own tests hold for the hot path: LPStateSpec.groovy and
LPSolverSpec.groovy, transcribed as data in tests/golden/reference_vectors.json.  All of these vectors are
small integers / dyadic fractions, so the fp64 instantiation must reproduce them exactly too."""
import numpy as np
import pytest

STATUS = {"OPTIMAL": 0, "UNBOUNDED": 1, "INFEASIBLE": 2}
KINDS = [0, 1, 2]  # DEC15, FP64, FP64_FUSED (every vector here is exact in all three)


@pytest.mark.parametrize("kind", KINDS)
def test_get_entering(oracle, reference_vectors, kind):        # LPStateSpec.groovy:12-29
    for case in reference_vectors["get_entering"]["cases"]:
        c = case["c"]
        st = oracle.State(np.zeros((0, len(c))), [], c, kind=kind)
        assert st.get_entering() == case["entering"], case


@pytest.mark.parametrize("kind", KINDS)
def test_get_leaving(oracle, reference_vectors, kind):         # LPStateSpec.groovy:31-48
    g = reference_vectors["get_leaving"]
    st = oracle.State(g["A"], g["b"], [0, 0, 0, 0], kind=kind)
    for case in g["cases"]:
        assert st.get_leaving(case["entering"]) == case["leaving"], case
    with pytest.raises(ValueError):                            # Validate.isTrue, LPState.java:288
        st.get_leaving(4)
    with pytest.raises(ValueError):
        st.get_leaving(-1)


@pytest.mark.parametrize("kind", KINDS)
def test_pivot_vectors(oracle, reference_vectors, kind):       # LPStateSpec.groovy:50-163
    for group in reference_vectors["pivot"]:
        for case in group["cases"]:
            st = oracle.State(group["A"], group["b"], group["c"], perm=group["perm"], kind=kind)
            threads = 4 if group["concurrent"] else 1          # pivotConcurrently: THREAD_AMOUNT = 4
            assert st.pivot(case["entering"], case["leaving"], threads=threads) == 0
            A, b, c, v, perm = st.read()
            assert np.array_equal(A, np.array(case["resA"], dtype=float)), (group["source"], case)
            assert np.array_equal(b, np.array(case["resB"], dtype=float))
            assert np.array_equal(c, np.array(case["resC"], dtype=float))
            assert v == case["resV"]
            assert list(perm) == case["resPerm"]


@pytest.mark.parametrize("kind", KINDS)
def test_sequential_and_concurrent_pivot_agree(oracle, reference_vectors, kind):
    g = reference_vectors["pivot"][2]
    for case in g["cases"]:
        dumps = []
        for threads in (1, 2, 4):
            st = oracle.State(g["A"], g["b"], g["c"], perm=g["perm"], kind=kind)
            st.pivot(case["entering"], case["leaving"], threads=threads)
            dumps.append(st.dump())
        assert dumps[0] == dumps[1] == dumps[2]


@pytest.mark.parametrize("kind", KINDS)
def test_min_in_b(oracle, reference_vectors, kind):            # LPSolverSpec.groovy:8-21
    for case in reference_vectors["min_in_b"]["cases"]:
        assert oracle.min_in_b(case["b"], kind=kind) == case["answer"], case


@pytest.mark.parametrize("kind", KINDS)
def test_conversion_into_aux_lp(oracle, reference_vectors, kind):   # LPSolverSpec.groovy:37-57
    g = reference_vectors["aux_lp_conversion"]
    st = oracle.convert_into_aux_lp(g["A"], g["b"], kind=kind)
    A, b, c, v, perm = st.read()
    assert np.array_equal(A, np.array(g["resA"], dtype=float))
    assert np.array_equal(c, np.array(g["resC"], dtype=float))
    assert np.array_equal(b, np.array(g["b"], dtype=float))
    n, m = 5, 5
    assert perm[n] == n + m and sorted(perm) == list(range(n + m + 1))   # x0 present exactly once


@pytest.mark.parametrize("kind", KINDS)
def test_solve_vectors(oracle, reference_vectors, kind):       # LPSolverSpec.groovy:76-111, :151-192
    for case in reference_vectors["solve"]:
        res, st = oracle.solve(case["A"], case["b"], case["c"], maximize=case["maximize"], kind=kind,
                               restore_order=case.get("restore_order"))
        assert res["status"] == STATUS[case["status"]], (case["source"], res)
        if case["status"] == "OPTIMAL":
            assert float(res["objective_text"]) == case["answer"], (case["source"], res)
            # the unrounded v need not be exact (decimal-15 gives -16.9999999999999 for the `min`
            # vector); the reference only ever exposes setScale(6, HALF_UP)  (LPSolver.java:113)
            assert abs(res["objective"] - case["answer"]) < 1e-9


@pytest.mark.parametrize("kind", KINDS)
def test_initial_infeasible_x0_final_slot(oracle, reference_vectors, kind):
    # the reference's own log shows x0 ending in slot 1 for this LP (logs/lp_solver.log:196)
    case = reference_vectors["solve"][2]
    res, _ = oracle.solve(case["A"], case["b"], case["c"], maximize=True, kind=kind, restore_order=[0, 1])
    assert res["phase1_used"] and res["x0_slot"] == 1


@pytest.mark.parametrize("kind", KINDS)
def test_solve_aux_lp(oracle, reference_vectors, kind):        # LPSolverSpec.groovy:113-124
    g = reference_vectors["solve_aux_lp"]
    st = oracle.State(g["A"], g["b"], g["c"], kind=kind)
    oracle.solve_aux_lp(st, g["index_of_x0"], g["min_in_b"])
    assert st.read()[3] == g["resV"]


@pytest.mark.parametrize("kind", KINDS)
def test_restore_initial_lp(oracle, reference_vectors, kind):  # LPSolverSpec.groovy:126-149
    g = reference_vectors["restore_initial_lp"]
    # ids: x1,x2 -> 0,1 (originals), x3..x6 -> 2..5 (slacks), x0 -> 6 (= n+m)
    ident = {"x1": 0, "x2": 1, "x3": 2, "x4": 3, "x5": 4, "x6": 5, "x0": 6}
    perm = [ident[s] for s in g["aux_names"]]
    aux = oracle.State(g["auxA"], g["auxB"], g["auxC"], perm=perm, kind=kind)
    status, st = oracle.restore_initial_lp(aux, g["initial_c"], g["index_of_x0"], [0, 1])
    assert status == 0
    A, b, c, v, p = st.read()
    assert np.array_equal(A, np.array(g["resA"], dtype=float))
    assert np.array_equal(b, np.array(g["resB"], dtype=float))
    assert np.array_equal(c, np.array(g["resC"], dtype=float))
    assert v == g["resV"]
    assert list(p) == [ident[s] for s in g["res_names"]]
