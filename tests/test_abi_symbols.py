"""CPU-side checks of the drop-in boundary: liblpx.so loads without a GPU and exports every symbol that
include/lpx.h declares; host-only entry points behave; the product package never imports the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lpxlib():
    import __graft_entry__ as g
    from linear_programming_solver_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lpx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lpxlib):
    L = lpxlib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    bound = {name for name, _, _ in lpxlib.SYMBOLS}
    for name in names:
        assert hasattr(L, name), "liblpx.so does not export " + name
        assert name in bound, "python binding misses " + name
    assert L.lpx_abi_version() == 5


def test_status_messages_match_reference_text(lpxlib):
    # LPSolver.java:105, :173, :149, :193
    assert lpxlib.status_message(lpxlib.UNBOUNDED) == "This linear program is unbounded"
    assert lpxlib.status_message(lpxlib.INFEASIBLE) == "This linear program is infeasible"
    assert lpxlib.status_message(lpxlib.AUX_UNBOUNDED) == "Auxiliary lp is unbounded"
    assert lpxlib.status_message(lpxlib.NO_DEGENERATE_PIVOT) == "Can't perform degenerate pivot"
    assert lpxlib.status_message(lpxlib.OPTIMAL) == ""


def test_java_default_name_order_matches_oracle_and_generator(lpxlib, oracle, decimal_goldens):
    L = lpxlib.lib()
    for n_str, want in decimal_goldens["java_default_name_order"].items():
        n = int(n_str)
        out = np.zeros(n, dtype=np.int32)
        assert L.lpx_java_default_name_order(n, out.ctypes.data_as(lpxlib.ip)) == 0
        assert out.tolist() == want
    out = np.zeros(4096, dtype=np.int32)
    L.lpx_java_default_name_order(4096, out.ctypes.data_as(lpxlib.ip))
    assert out.tolist() == oracle.java_default_name_order(4096).tolist()


def test_hashmap_key_order_python_mirror(decimal_goldens):
    from linear_programming_solver_amd.java_compat import hashmap_key_order, java_string_hash
    assert java_string_hash("x1") == 3769 and java_string_hash("x10") == 116887   # "x".hashCode()*31 + ...
    for n_str, want in decimal_goldens["java_default_name_order"].items():
        n = int(n_str)
        names = ["x%d" % (i + 1) for i in range(n)]
        assert [int(k[1:]) - 1 for k in hashmap_key_order(names)] == want


def test_errors_without_gpu_are_loud(lpxlib):
    """No GPU in this container: creating a state must fail with a device error, never fall back."""
    L = lpxlib.lib()
    if L.lpx_device_count() > 0:
        pytest.skip("a GPU is visible")
    from linear_programming_solver_amd import LPState
    with pytest.raises(RuntimeError):
        LPState([[1.0]], [1.0], [1.0])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "linear_programming_solver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "liblporacle" not in text and "lp_oracle" not in text, f


def test_header_is_valid_c99_and_cxx(tmp_path):
    """include/lpx.h is the drop-in boundary: it must compile as plain C (cgo/JNI glue) and as C++."""
    import subprocess
    src = tmp_path / "use_lpx.c"
    src.write_text('#include "lpx.h"\nint main(void){ lpx_solve_result r; (void)r; return LPX_OPTIMAL + (int)sizeof(lpx_solve_options) * 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])


def test_python_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors of lpx_solve_result / lpx_solve_options must have the C sizes and offsets."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "lpx.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(lpx_solve_result), offsetof(lpx_solve_result, objective_text),
         offsetof(lpx_solve_result, pivots_phase1), offsetof(lpx_solve_result, seconds_pivots),
         sizeof(lpx_solve_options), offsetof(lpx_solve_options, keep_state),
         offsetof(lpx_solve_options, restore_order_len), sizeof(lpx_state_info), offsetof(lpx_state_info, sweep_rows));
  return 0; }
''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from linear_programming_solver_amd import _lib
    R, O, I = _lib.SolveResult, _lib.SolveOptions, _lib.StateInfo
    assert got == [C.sizeof(R), R.objective_text.offset, R.pivots_phase1.offset, R.seconds_pivots.offset,
                   C.sizeof(O), O.keep_state.offset, O.restore_order_len.offset, C.sizeof(I), I.sweep_rows.offset]


def test_option_keys_match_the_header():
    """_lib.OPTIONS mirrors enum lpx_option."""
    from linear_programming_solver_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lpx.h")).read(), flags=re.S)
    enum = dict((name.lower(), int(val)) for name, val in re.findall(r"LPX_OPT_([A-Z0-9_]+) = (\d+)", text))
    count = enum.pop("count")
    assert enum == _lib.OPTIONS and count == len(enum)
