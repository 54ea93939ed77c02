"""Pins the oracle's THIRD instantiation (oracle/lp_oracle.hpp Num<F64Fused>: IEEE fp64 with every update x - c*r /
x + a*b as ONE fused multiply-add — the checker of the GPU's opt-in fused-arithmetic mode, LPX_OPT_FUSED) against an
independent restatement in exact rational arithmetic: the reference's pivot (LPState.java:133-181) with every
operation computed exactly on fractions.Fraction and rounded ONCE to the nearest double (int / int true division in
Python is correctly rounded), the product-and-difference of :162 / :164 / :177 / :171 as one such operation."""
from fractions import Fraction

import numpy as np
import pytest


def _rn(q):
    """nearest double of an exact rational (ties to even: CPython's long true division is correctly rounded)."""
    return q.numerator / q.denominator


def _fma(a, b, c):
    r = _rn(Fraction(a) * Fraction(b) + Fraction(c))
    if r == 0.0:   # exact zero: IEEE gives +0 unless both addends are -0 (round to nearest)
        prod_neg = (np.signbit(a) != np.signbit(b))
        return -0.0 if (prod_neg and np.signbit(c)) else 0.0
    return r


def _div(a, b):
    r = _rn(Fraction(a) / Fraction(b))
    if r == 0.0 and a == 0.0:
        return -0.0 if (np.signbit(a) != np.signbit(b)) else 0.0
    return r


def pivot_fused_py(A, b, c, v, e, l):
    """LPState.pivotSequentially (LPState.java:133-181), fused updates, in pure Python."""
    m, n = A.shape
    A = A.copy(); b = b.copy(); c = c.copy()
    p = A[l, e]
    A[l, e] = _div(1.0, p)                                      # :139
    for j in range(n):
        if j != e:
            A[l, j] = _div(A[l, j], p)                          # :144
    b[l] = _div(b[l], p)                                        # :146
    for i in range(m):
        if i == l:
            continue
        ce = A[i, e]
        A[i, e] = -_div(ce, p)                                  # :157
        for j in range(n):
            if j != e:
                A[i, j] = _fma(-ce, A[l, j], A[i, j])           # :162 as one operation
        b[i] = _fma(-ce, b[l], b[i])                            # :164
    pc = c[e]
    v = _fma(b[l], pc, v)                                       # :171
    c[e] = -_div(pc, p)                                         # :172
    for j in range(n):
        if j != e:
            c[j] = _fma(-pc, A[l, j], c[j])                     # :177
    return A, b, c, v


def _bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fused_pivots_match_exact_rational_restatement(oracle, seed):
    rng = np.random.default_rng(seed)
    m, n = 7, 9
    A = rng.random((m, n)) - 0.2
    b = 1.0 + rng.random(m)
    c = rng.random(n) - 0.3
    st = oracle.State(A, b, c, kind=oracle.FP64_FUSED)
    v = 0.0
    for _ in range(6):
        e = st.get_entering()
        if e < 0:
            break
        l = st.get_leaving(e)
        if l < 0:
            break
        assert st.pivot(e, l) == 0
        A, b, c, v = pivot_fused_py(A, b, c, v, e, l)
        gA, gb, gc, gv, _ = st.read()
        assert np.array_equal(_bits(gA), _bits(A))
        assert np.array_equal(_bits(gb), _bits(b))
        assert np.array_equal(_bits(gc), _bits(c))
        assert _bits([gv])[0] == _bits([v])[0]


def test_fused_differs_from_unfused_in_bits_but_not_in_outcome(oracle):
    rng = np.random.default_rng(11)
    m, n = 48, 80
    A = rng.random((m, n)); b = (n / 4.0) * (1.0 + rng.random(m)); c = rng.random(n)
    r1, s1 = oracle.solve(A, b, c, True, kind=oracle.FP64)
    r2, s2 = oracle.solve(A, b, c, True, kind=oracle.FP64_FUSED)
    r0, s0 = oracle.solve(A, b, c, True, kind=oracle.DEC15)
    assert r1["status"] == r2["status"] == r0["status"] == 0
    # same pivot sequence and basis as the reference's decimal arithmetic, objective within the stated tolerance
    assert np.array_equal(r2["trace"], r0["trace"]) and np.array_equal(r1["trace"], r0["trace"])
    assert list(s2.read()[4]) == list(s0.read()[4])
    assert abs(r2["objective"] - r0["objective"]) <= 1e-9 * max(1.0, abs(r0["objective"]))
    assert r2["objective_text"] == r0["objective_text"]
    # ... while the two binary instantiations are different arithmetic (so the fused GPU mode needs its own checker)
    assert not np.array_equal(_bits(s1.read()[0]), _bits(s2.read()[0]))


def test_fused_identity_step_keeps_every_value(oracle):
    """A multiplier of +0 must leave x unchanged bit for bit (-0.0 included): the sweeps of a partly filled block pad
    with such steps (fma(-(+0), +0, x) = x + (-0) = x)."""
    A = np.array([[1.0, -0.0, 3.5], [0.0, 2.0, -0.0]])
    b = np.array([1.0, -0.0])
    c = np.array([1.0, 0.0, 0.0])
    st = oracle.State(A, b, c, kind=oracle.FP64_FUSED)
    assert st.pivot(0, 0) == 0            # row 1 has multiplier A[1][0] = +0
    gA, gb, _, _, _ = st.read()
    assert np.array_equal(_bits(gA[1, 1:]), _bits(A[1, 1:]))
    assert _bits(gb)[1] == _bits(b)[1]
