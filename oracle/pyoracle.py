"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/liblporacle.so, the CPU restatement of the reference's simplex hot path
(see oracle/lp_oracle.hpp for the reference file:line each routine follows).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker / baseline — never the product path (linear_programming_solver_amd/ must not import it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblporacle.so")

DEC15 = 0  # decimal-15 HALF_UP: the reference's BigDecimal/MathContext(15, HALF_UP) semantics
FP64 = 1   # IEEE double, unfused: what the HIP kernels compute
FP64_FUSED = 2  # IEEE double with x - c*r as ONE fused multiply-add: the checker of the opt-in fused mode (LPX_OPT_FUSED)


class OrcResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("phase1_used", C.c_int32),
        ("objective", C.c_double),
        ("objective_repr", C.c_char * 64),
        ("objective_text", C.c_char * 64),
        ("pivots1", C.c_int64),
        ("pivots2", C.c_int64),
        ("x0_slot", C.c_int32),
        ("final_m", C.c_int32),
        ("final_n", C.c_int32),
        ("trace_len", C.c_int32),
        ("seconds", C.c_double),
    ]


def build(force=False):
    """Compile the oracle with gcc (host only)."""
    srcs = [os.path.join(_HERE, f) for f in ("lp_oracle.cpp", "lp_oracle.hpp", "dec15.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "lpx.h"))
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liblporacle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    try:   # the library may have been built with -mfma on another machine (it travels with the repository snapshot)
        if " fma " not in " " + open("/proc/cpuinfo").read().replace("\n", " ") + " ":
            subprocess.check_call(["make", "-C", _HERE, "-B", "liblporacle.so", "FMA_FLAG="], stdout=subprocess.DEVNULL)
    except OSError:
        pass
    L = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    L.orc_dec_op.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.orc_dec_op.restype = C.c_int
    L.orc_round6_double.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    L.orc_state_new.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_double, ip, C.c_int]
    L.orc_state_new.restype = C.c_void_p
    L.orc_state_free.argtypes = [C.c_void_p]
    L.orc_state_free.restype = None
    L.orc_get_entering.argtypes = [C.c_void_p]
    L.orc_get_leaving.argtypes = [C.c_void_p, C.c_int]
    L.orc_pivot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_state_dims.argtypes = [C.c_void_p, ip, ip, ip]
    L.orc_state_dims.restype = None
    L.orc_state_read.argtypes = [C.c_void_p, dp, dp, dp, dp, ip]
    L.orc_state_read.restype = None
    L.orc_state_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.orc_state_dump.restype = C.c_int64
    L.orc_simplex_loop.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int64), ip, ip, C.c_int64]
    L.orc_simplex_loop.restype = C.c_double
    L.orc_solve.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_int, ip, C.c_int64, C.c_int,
                            C.POINTER(OrcResult), ip, C.c_int64]
    L.orc_solve.restype = C.c_void_p
    L.orc_solve2.argtypes = L.orc_solve.argtypes + [C.c_int]
    L.orc_solve2.restype = C.c_void_p
    L.orc_state_set_pricing.argtypes = [C.c_void_p, C.c_int]
    L.orc_state_set_pricing.restype = None
    L.orc_min_in_b.argtypes = [C.c_int, C.c_int, dp]
    L.orc_solve_aux_lp.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_convert_into_aux_lp.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp]
    L.orc_convert_into_aux_lp.restype = C.c_void_p
    L.orc_restore_initial_lp.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, ip, ip]
    L.orc_restore_initial_lp.restype = C.c_void_p
    L.orc_java_default_name_order.argtypes = [C.c_int, ip]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def dec_op(op, a, b=""):
    """op in add/sub/mul/div/cmp/norm/scale6 on decimal strings -> canonical string (None: div by zero)."""
    code = {"add": 0, "sub": 1, "mul": 2, "div": 3, "cmp": 4, "norm": 5, "scale6": 6}[op]
    buf = C.create_string_buffer(128)
    rc = lib().orc_dec_op(code, str(a).encode(), str(b).encode(), buf, 128)
    if rc == 1:
        return None
    if rc != 0:
        raise RuntimeError("orc_dec_op failed: %d" % rc)
    return buf.value.decode()


def round6_double(v):
    buf = C.create_string_buffer(1400)
    lib().orc_round6_double(float(v), buf, 1400)
    return buf.value.decode()


def min_in_b(b, kind=DEC15):
    b = _f64(b)
    return lib().orc_min_in_b(kind, b.size, _dp(b) if b.size else None)


def java_default_name_order(n):
    out = np.zeros(max(n, 1), dtype=np.int32)
    lib().orc_java_default_name_order(n, _ip(out))
    return out[:n].copy()


class State:
    """Mirror of the reference's LPState over the oracle (LPState.java)."""

    def __init__(self, A, b, c, v=0.0, perm=None, kind=DEC15, with_perm=True, _handle=None, pricing=0):
        self.kind = kind
        if _handle is not None:
            self._h = _handle
        else:
            b = _f64(b)
            c = _f64(c)
            m, n = b.size, c.size
            A = _f64(A, (m, n)) if m * n else np.zeros((m, n))
            p = None if perm is None else np.ascontiguousarray(np.asarray(perm, dtype=np.int32))
            self._h = lib().orc_state_new(kind, m, n, _dp(A) if A.size else None, _dp(b) if m else None,
                                          _dp(c) if n else None, float(v), None if p is None else _ip(p),
                                          1 if with_perm else 0)
        m_, n_, hp = C.c_int32(), C.c_int32(), C.c_int32()
        lib().orc_state_dims(self._h, C.byref(m_), C.byref(n_), C.byref(hp))
        self.m, self.n, self.has_perm = m_.value, n_.value, bool(hp.value)
        if pricing:
            lib().orc_state_set_pricing(self._h, int(pricing))

    def close(self):
        if self._h:
            lib().orc_state_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_entering(self):
        return lib().orc_get_entering(self._h)

    def get_leaving(self, e):
        r = lib().orc_get_leaving(self._h, int(e))
        if r == -2:
            raise ValueError("IllegalArgumentException")
        return r

    def pivot(self, e, l, threads=1):
        return lib().orc_pivot(self._h, int(e), int(l), int(threads))

    def read(self):
        A = np.zeros((self.m, self.n))
        b = np.zeros(self.m)
        c = np.zeros(self.n)
        v = C.c_double()
        perm = np.zeros(self.n + self.m, dtype=np.int32)
        lib().orc_state_read(self._h, _dp(A) if A.size else None, _dp(b) if self.m else None,
                             _dp(c) if self.n else None, C.byref(v), _ip(perm) if self.has_perm else None)
        return A, b, c, v.value, (perm if self.has_perm else None)

    def dump(self):
        """Canonical text of every entry (decimal 'ce' form / C99 %a), for bit-for-bit comparison."""
        need = lib().orc_state_dump(self._h, None, 0)
        buf = C.create_string_buffer(int(need))
        lib().orc_state_dump(self._h, buf, need)
        out = {}
        for line in buf.value.decode().splitlines():
            k, *vals = line.split(" ")
            out[k] = vals
        return out

    def simplex_loop(self, max_pivots=-1, threads=1, want_trace=False, trace_cap=1 << 20):
        piv = C.c_int64()
        st = C.c_int32()
        tr = np.zeros((trace_cap if want_trace else 1, 2), dtype=np.int32)
        secs = lib().orc_simplex_loop(self._h, int(max_pivots), int(threads), C.byref(piv), C.byref(st),
                                      _ip(tr) if want_trace else None, trace_cap)
        return {"pivots": piv.value, "status": st.value, "seconds": secs,
                "trace": tr[: piv.value].copy() if want_trace else None}


def convert_into_aux_lp(A, b, kind=DEC15):
    """LPSolver.convertIntoAuxLP (LPSolver.java:283): State of the auxiliary LP."""
    b = _f64(b)
    m = b.size
    A = _f64(A, (m, -1))
    h = lib().orc_convert_into_aux_lp(kind, m, A.shape[1], _dp(A), _dp(b))
    return State(None, None, None, kind=kind, _handle=h)


def solve_aux_lp(state, index_of_x0, mib):
    """LPSolver.solveAuxLP (LPSolver.java:135): returns x0's final slot (raises on an unbounded aux LP)."""
    r = lib().orc_solve_aux_lp(state._h, int(index_of_x0), int(mib))
    if r <= -1000:
        raise RuntimeError("aux lp status %d" % (-1000 - r))
    return r


def restore_initial_lp(aux_state, initial_c, x0_slot, order):
    """LPSolver.restoreInitialLP (LPSolver.java:200): (status, new State | None)."""
    c0 = _f64(initial_c)
    order = np.ascontiguousarray(np.asarray(order, dtype=np.int32))
    st = C.c_int32()
    h = lib().orc_restore_initial_lp(aux_state._h, _dp(c0), c0.size, int(x0_slot), _ip(order), C.byref(st))
    if not h:
        return st.value, None
    return st.value, State(None, None, None, kind=aux_state.kind, _handle=h)


def solve(A, b, c, maximize=True, kind=DEC15, restore_order=None, max_pivots=-1, threads=1,
          want_trace=True, trace_cap=1 << 20, pricing=0):
    """LPSolver.solve (LPSolver.java:78).  Returns (result dict, final State)."""
    b = _f64(b)
    c = _f64(c)
    m, n = b.size, c.size
    A = _f64(A, (m, n))
    ro = None if restore_order is None else np.ascontiguousarray(np.asarray(restore_order, dtype=np.int32))
    res = OrcResult()
    tr = np.zeros((trace_cap if want_trace else 1, 3), dtype=np.int32)
    h = lib().orc_solve2(kind, m, n, _dp(A), _dp(b), _dp(c), 1 if maximize else 0,
                         None if ro is None else _ip(ro), int(max_pivots), int(threads), C.byref(res),
                         _ip(tr) if want_trace else None, trace_cap, int(pricing))
    st = State(None, None, None, kind=kind, _handle=h)
    out = {
        "status": res.status, "phase1_used": bool(res.phase1_used), "objective": res.objective,
        "objective_repr": res.objective_repr.decode(), "objective_text": res.objective_text.decode(),
        "pivots1": res.pivots1, "pivots2": res.pivots2, "x0_slot": res.x0_slot,
        "seconds": res.seconds,
        "trace": tr[: min(res.trace_len, trace_cap)].copy() if want_trace else None,
    }
    return out, st
