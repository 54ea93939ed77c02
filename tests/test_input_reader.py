"""LPInputReader (host-side text -> LPStandardForm plumbing) against the reference's own
expectations: LPInputReaderSpec.groovy:7-52, LPInputReaderTest.java:28-179 and io_files/input.txt:1-16."""
import os
from decimal import Decimal

import numpy as np
import pytest

from linear_programming_solver_amd.errors import LPException
from linear_programming_solver_amd.lp_input_reader import LPInputReader


def test_simple_lp(reference_vectors):                           # LPInputReaderSpec.groovy:7-25
    g = reference_vectors["reader"][0]
    f = LPInputReader().read_lp(g["text"])
    assert f.A.tolist() == g["A"] and f.b.tolist() == g["b"] and f.c.tolist() == g["c"]
    assert f.variables == {i: nm for i, nm in enumerate(g["names"])}
    assert f.coefficients == {nm: i for i, nm in enumerate(g["names"])}
    assert (f.m, f.n, f.maximize) == (g["m"], g["n"], g["maximize"])


def test_complicated_lp_is_parsed_exactly(reference_vectors):     # LPInputReaderSpec.groovy:27-50
    g = reference_vectors["reader"][1]
    f = LPInputReader().read_lp(g["text"])
    assert (f.m, f.n, f.maximize) == (g["m"], g["n"], g["maximize"])
    assert f.exact["A"] == [[Decimal(x) for x in row] for row in g["A_text"]]
    assert f.exact["b"] == [Decimal(x) for x in g["b_text"]]        # 40-digit numbers survive, >= rows negated
    assert f.exact["c"] == [Decimal(x) for x in g["c_text"]]
    assert f.variables == {0: "x1", 1: "x2"}
    assert f.A[0, 0] == -6.338203729 and f.A[2, 1] == -2332.33214   # fp64 view handed to the device


def test_io_files_first_block(reference_vectors, tmp_path):       # io_files/input.txt:1-16 through readLP(File)
    g = reference_vectors["io_files_first_block"]
    path = tmp_path / "input.txt"
    path.write_text(g["text"] + "\nmax\nx1+2x2+3x3+x4+x5\nx1+x2+4x3+-x4+x5=1\n")   # a second block follows
    f = LPInputReader().read_lp(str(path))
    assert (f.m, f.n, f.maximize) == (14, 18, True)                # only the first block is consumed (:76-81)
    assert f.c.tolist() == [1.0] * 18 and f.b.tolist() == [1.0] * 14
    assert f.A.sum() == 36 and f.A[9].nonzero()[0].tolist() == [1, 4, 8, 10]        # x2 + x5 + x9 + x11 <= 1


def test_late_variables_and_padding():                             # LPInputReader.java:172-178, :215-223
    f = LPInputReader().read_lp("max\nx1 + 2x2\nx1 + x3 <= 4\n-y - x2 >= -7\nx1 = 2")
    assert f.variables == {0: "x1", 1: "x2", 2: "x3", 3: "y"}
    assert f.c.tolist() == [1, 2, 0, 0]
    assert f.A.tolist() == [[1, 0, 1, 0], [0, 1, 0, 1], [1, 0, 0, 0], [-1, 0, 0, 0]]
    assert f.b.tolist() == [4, 7, 2, -2]
    assert f.m == 4 and f.n == 4


def test_coefficient_forms():
    f = LPInputReader().read_lp("min\n-x + 0.5*y - 3z\n2.5x - y <= 1\n x+y+z == 3 \n- x >= - 4.25")
    assert not f.maximize and f.c.tolist() == [-1, 0.5, -3]
    assert f.A.tolist() == [[2.5, -1, 0], [1, 1, 1], [-1, -1, -1], [1, 0, 0]]
    assert f.b.tolist() == [1, 3, -3, 4.25]


@pytest.mark.parametrize("text,msg", [
    ("maximize\nx1\nx1 <= 1", "Incorrect max/min parameter"),       # LPInputReaderTest.java
    ("max\nx1 + \nx1 <= 1", "Can't recognize objective"),
    ("max\n3 + x1\nx1 <= 1", "Can't recognize objective"),
    ("max\nx1\nx1 < 1", "Can't recognize constraint"),
    ("max\nx1\nx1 + x2", "Can't recognize constraint"),
    ("max\nx1+2x2+3x3+x4+x5\nx1+x2+4x3+-x4+x5=1\nx1<=1", "Can't recognize constraint"),   # io_files/input.txt:18-22 ('+-x4')
    ("max\nx1", "Incomplete lp"),
    ("", "Incomplete lp"),
])
def test_error_messages(text, msg):
    with pytest.raises(LPException) as ei:
        LPInputReader().read_lp_string(text)
    assert str(ei.value) == msg


def test_file_errors(tmp_path):
    with pytest.raises(ValueError):                                # not a file -> IllegalArgumentException (:54-57)
        LPInputReader().read_lp_file(str(tmp_path))
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    with pytest.raises(LPException) as ei:
        LPInputReader().read_lp_file(str(empty))
    assert str(ei.value) == "Input file is empty"
    noc = tmp_path / "noc.txt"
    noc.write_text("max\nx1 + x2\n\nx1 <= 1\n")
    with pytest.raises(LPException) as ei:
        LPInputReader().read_lp_file(str(noc))
    assert str(ei.value) == "No constraints in the input file"


def test_get_dual_metadata_only(reference_vectors):
    # the transpose itself runs on the device (tests/test_gpu_parity.py); here: shapes on an empty form
    from linear_programming_solver_amd.lp_standard_form import LPStandardForm
    f = LPStandardForm(np.zeros((0, 3)), [], [1, 2, 3], maximize=True)
    assert f.m == 0 and f.n == 3 and not f.has_variable_names()


# ---- host-side helpers of LPSolver (no device work) ------------------------------------------------------
def test_min_in_b_vectors(reference_vectors):                     # LPSolverSpec.groovy:8-21
    from linear_programming_solver_amd.lp_solver import LPSolver
    for case in reference_vectors["min_in_b"]["cases"]:
        assert LPSolver.min_in_b(case["b"]) == case["answer"], case


def test_name_for_x0():                                           # LPSolverSpec.groovy:23-35
    from linear_programming_solver_amd.lp_solver import LPSolver
    assert LPSolver.get_name_for_x0({"x1": 0, "x2": 1, "x3": 2}) == "x0"
    assert LPSolver.get_name_for_x0({"x0": 0, "x1": 1}) == "auxVar"
    assert LPSolver.get_name_for_x0({"auxVar": 0, "x0": 1, "auxVar1": 2}) == "auxVar2"
    for coeffs in ({"x1": 0}, {"x0": 0, "x1": 1}, {"auxVar": 0, "x0": 1, "auxVar1": 2}):
        assert LPSolver.get_name_for_x0(coeffs) not in coeffs


def test_slack_names_skip_used_names():                           # LPSolverSpec.groovy:59-74 (x0,x1,x4,x6 taken)
    from linear_programming_solver_amd.lp_solver import LPSolver
    assert LPSolver.slack_names({"x0": 0, "x1": 1, "x4": 2, "x6": 3}, 4) == ["x2", "x3", "x5", "x7"]
