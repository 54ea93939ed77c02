#!/usr/bin/env python3
"""Generates tests/golden/decimal_goldens.json.

The reference (Java, BigDecimal + MathContext(15, HALF_UP)) cannot be run in the build image: there is no
JVM.  Python's stdlib `decimal` implements the same General Decimal Arithmetic rule
("exact result, rounded once to `prec` significant digits"), so this script is an INDEPENDENT second
restatement of the reference's hot path, written directly from the Java sources' operation order:

    LPState.java:133-181  pivotSequentially      LPState.java:274-305  getEntering / getLeaving
    LPSolver.java:78-246  solve .. restoreInitialLP      LPSolver.java:283-321 convertIntoAuxLP

Its outputs (pivot traces, full tableaux as canonical coefficient/exponent text, objective text) pin
oracle/dec15.hpp + oracle/lp_oracle.hpp bit for bit (tests/test_oracle_goldens.py); the reference's own
Spock expectations are transcribed as data in reference_vectors.json and checked against both.

Run:  python tests/golden/gen_golden.py      (deterministic; rewrites decimal_goldens.json)
"""
import json
import os
import random
from decimal import ROUND_HALF_UP, Context, Decimal, DivisionByZero

CTX = Context(prec=15, rounding=ROUND_HALF_UP, Emax=999999, Emin=-999999)
EPS = Decimal("1e-9")    # LPState.java:20
INF = Decimal("1e50")    # LPState.java:21
ZERO = Decimal(0)
ONE = Decimal(1)

OPTIMAL, UNBOUNDED, INFEASIBLE, AUX_UNBOUNDED, NO_DEGENERATE_PIVOT, BAD_ARGUMENT, RESTORE_INDEX_FAULT = range(7)
DIVIDE_BY_ZERO = 8


def canon(d):
    """Canonical text: integer coefficient without trailing zeros + 'e' + exponent; '0' for zero."""
    if d == 0:
        return "0"
    sign, digits, exp = d.as_tuple()
    digits = list(digits)
    while digits and digits[-1] == 0:
        digits.pop()
        exp += 1
    return ("-" if sign else "") + "".join(map(str, digits)) + "e" + str(exp)


class LPState:
    def __init__(self, A, b, c, v=ZERO, perm=None):
        self.A = [list(r) for r in A]
        self.b = list(b)
        self.c = list(c)
        self.v = v
        self.m = len(self.b)
        self.n = len(self.c)
        self.perm = None if perm is None else list(perm)

    def get_entering(self):                                   # LPState.java:274-285
        for i in range(self.n):
            if self.c[i].compare(EPS) > 0:
                return i
        return -1

    def get_leaving(self, e):                                 # LPState.java:287-305
        assert 0 <= e < self.n
        leaving, min_slack = -1, INF
        for i in range(self.m):
            aie = self.A[i][e]
            slack = INF if aie.compare(EPS) < 0 else CTX.divide(self.b[i], aie)
            if slack.compare(min_slack) < 0:
                min_slack, leaving = slack, i
        return leaving

    def pivot(self, e, l):                                    # LPState.java:133-181
        A, b, c, n, m = self.A, self.b, self.c, self.n, self.m
        prow = A[l]
        piv = prow[e]
        prow[e] = CTX.divide(ONE, piv)
        for i in range(n):
            if i != e:
                prow[i] = CTX.divide(prow[i], piv)
        b[l] = CTX.divide(b[l], piv)
        bent = b[l]
        for i in range(m):
            if i == l:
                continue
            row = A[i]
            ce = row[e]
            row[e] = CTX.divide(ce, piv).copy_negate()
            for j in range(n):
                if j != e:
                    row[j] = CTX.subtract(row[j], CTX.multiply(ce, prow[j]))
            b[i] = CTX.subtract(b[i], CTX.multiply(ce, bent))
        pc = c[e]
        self.v = CTX.add(self.v, CTX.multiply(b[l], pc))
        c[e] = CTX.divide(pc, piv).copy_negate()
        for i in range(n):
            if i != e:
                c[i] = CTX.subtract(c[i], CTX.multiply(pc, prow[i]))
        if self.perm is not None:                             # exchangeIndexes :311-320
            self.perm[e], self.perm[l + n] = self.perm[l + n], self.perm[e]

    def dump(self):
        return {"A": [canon(x) for r in self.A for x in r], "b": [canon(x) for x in self.b],
                "c": [canon(x) for x in self.c], "v": canon(self.v), "perm": self.perm}


def min_in_b(b):                                              # LPSolver.java:375-386
    mn, idx = INF, -1
    for i, x in enumerate(b):
        if mn.compare(x) > 0:
            mn, idx = x, i
    return idx


def java_string_hash(s):
    h = 0
    for ch in s:
        h = (h * 31 + ord(ch)) & 0xFFFFFFFF
    return h


def java_default_name_order(n):
    """keySet() order of a default HashMap after put("x1") .. put("xn")  (LPSolver.java:388-400)."""
    cap = 16
    while n > 0.75 * cap:
        cap *= 2
    buckets = [[] for _ in range(cap)]
    for k in range(1, n + 1):
        h = java_string_hash("x%d" % k)
        h ^= h >> 16
        buckets[h & (cap - 1)].append(k - 1)
    assert max(len(bk) for bk in buckets) < 8
    return [i for bk in buckets for i in bk]


def simplex_loop(st, phase, trace, track=None, unbounded=UNBOUNDED):
    while True:
        e = st.get_entering()
        if e == -1:
            return OPTIMAL, track
        l = st.get_leaving(e)
        if l == -1:
            return unbounded, track
        if track is not None:                                 # LPSolver.java:151-155
            if e == track:
                track = l + st.n
            elif l + st.n == track:
                track = e
        st.pivot(e, l)
        trace.append([phase, e, l])


def solve(A, b, c, maximize, restore_order=None):
    """LPSolver.solve on copies (LPSolver.java:78-246).  Returns dict."""
    m, n = len(b), len(c)
    c0 = [x if maximize else x.copy_negate() for x in c]      # :86-89
    trace = []
    out = {"phase1_used": False, "x0_slot": -1}
    mib = min_in_b(b)
    if mib == -1 or b[mib].compare(ZERO) >= 0:
        st = LPState(A, b, c0, ZERO, list(range(n + m)))
    else:
        out["phase1_used"] = True
        na = n + 1
        auxA = [list(r) + [Decimal(-1)] for r in A]           # :283-321
        auxc = [ZERO] * n + [Decimal(-1)]
        perm = list(range(n)) + [n + m] + [n + i for i in range(m)]
        aux = LPState(auxA, b, auxc, ZERO, perm)
        aux.pivot(na - 1, mib)                                # :138
        trace.append([1, na - 1, mib])
        status, x0 = simplex_loop(aux, 1, trace, track=mib + na, unbounded=AUX_UNBOUNDED)
        out["x0_slot"] = x0
        if status != OPTIMAL:
            return dict(out, status=status, trace=trace, final=aux.dump(), v=aux.v)
        x0v = ZERO if x0 < na else aux.b[x0 - na]             # :169-174
        if abs(x0v).compare(EPS) > 0:
            return dict(out, status=INFEASIBLE, trace=trace, final=aux.dump(), v=aux.v)
        if x0 >= na:                                          # :182-198
            row = aux.A[x0 - na]
            ent = -1
            for i in range(na):
                if abs(row[i]).compare(EPS) > 0:
                    ent = i
                    break
            if ent == -1:
                return dict(out, status=NO_DEGENERATE_PIVOT, trace=trace, final=aux.dump(), v=aux.v)
            aux.pivot(ent, x0 - na)
            trace.append([1, ent, x0 - na])
            x0 = ent
        out["x0_slot"] = x0
        # restoreInitialLP :200-246
        An = [r[:x0] + r[x0 + 1:] for r in aux.A]
        slot_of = {var: s for s, var in enumerate(aux.perm)}
        order = restore_order if restore_order is not None else java_default_name_order(n)
        v = ZERO
        cn = [ZERO] * n
        for index in order:
            k = c0[index]
            cur = slot_of[index]
            if cur >= na:
                r = cur - na
                v = CTX.add(v, CTX.multiply(aux.b[r], k))
                for j in range(n):
                    cn[j] = CTX.add(cn[j], CTX.multiply(An[r][j].copy_negate(), k))
            else:
                if cur >= n:                                  # ArrayIndexOutOfBoundsException at :231
                    return dict(out, status=RESTORE_INDEX_FAULT, trace=trace, final=aux.dump(), v=aux.v)
                cn[cur] = CTX.add(cn[cur], k)                 # bug-for-bug: aux slot used as post-drop index
        perm = [p for s, p in enumerate(aux.perm) if s != x0]
        st = LPState(An, aux.b, cn, v, perm)
    status, _ = simplex_loop(st, 2, trace)
    return dict(out, status=status, trace=trace, final=st.dump(), v=st.v)


def scale6(v):
    """BigDecimal.setScale(6, HALF_UP).toString(); BigDecimal has no negative zero."""
    q = v.quantize(Decimal("0.000001"), rounding=ROUND_HALF_UP)
    if q == 0:
        q = q.copy_abs()
    return str(q)


# ------------------------------------------------------------------------------------------- generators
def rdec(rng, lo, hi, places):
    x = rng.uniform(lo, hi)
    return Decimal(("%." + str(places) + "f") % x)


def gen_dense_feasible(rng, m, n):
    A = [[rdec(rng, 0.01, 1, 6) for _ in range(n)] for _ in range(m)]
    b = [rdec(rng, n / 4.0, n / 2.0, 6) for _ in range(m)]
    c = [rdec(rng, 0.01, 1, 6) for _ in range(n)]
    return A, b, c, True


def gen_mixed(rng, m, n, neg_b=0.3, places=3, lo=-5, hi=5):
    A = [[rdec(rng, lo, hi, places) for _ in range(n)] for _ in range(m)]
    b = [rdec(rng, -3 if rng.random() < neg_b else 0.5, 10, places) for _ in range(m)]
    c = [rdec(rng, -2, 4, places) for _ in range(n)]
    return A, b, c, rng.random() < 0.7


def gen_integer_degenerate(rng, m, n):
    """Small-integer data with many ties and zeros (exact in both radices)."""
    A = [[Decimal(rng.choice([0, 0, 1, 1, 1, -1, 2])) for _ in range(n)] for _ in range(m)]
    b = [Decimal(rng.choice([0, 1, 1, 2, 2, -1])) for _ in range(m)]
    c = [Decimal(rng.choice([1, 1, 2, 0, -1])) for _ in range(n)]
    return A, b, c, True


def lp_case(name, A, b, c, maximize, keep_final=True):
    res = solve(A, b, c, maximize)
    v = res["v"] if maximize else res["v"].copy_negate()
    case = {
        "name": name, "m": len(b), "n": len(c), "maximize": maximize,
        "A": [str(x) for r in A for x in r], "b": [str(x) for x in b], "c": [str(x) for x in c],
        "status": res["status"], "phase1_used": res["phase1_used"], "x0_slot": res["x0_slot"],
        "trace": res["trace"], "objective_repr": canon(v), "objective_text": scale6(v),
    }
    if keep_final:
        case["final"] = res["final"]
    return case


def main():
    rng = random.Random(20261003)
    out = {"_comment": "generated by tests/golden/gen_golden.py with Python decimal (prec=15, ROUND_HALF_UP)"}

    # 1) scalar operation vectors -------------------------------------------------------------------
    ops = []

    def rnd_operand():
        kind = rng.random()
        digits = rng.randint(1, 15)
        coef = rng.randint(1, 10 ** digits - 1)
        if kind < 0.15:
            coef = rng.choice([1, 5, 10 ** (digits - 1), 10 ** digits - 1, 5 * 10 ** (digits - 1)])
        exp = rng.randint(-20, 12) if rng.random() < 0.8 else rng.randint(-60, 55)
        sign = "-" if rng.random() < 0.4 else ""
        return "%s%de%d" % (sign, coef, exp)

    for _ in range(3000):
        a, b = rnd_operand(), rnd_operand()
        if rng.random() < 0.25:  # near-cancellation / near-tie operands
            da = Decimal(a)
            b = str(CTX.add(da.copy_negate(), Decimal(rnd_operand()) * Decimal("1e-12")))
            if rng.random() < 0.5:
                b = str(Decimal(b).copy_negate())
        da, db = Decimal(a), Decimal(b)
        rec = {"a": a, "b": b, "add": canon(CTX.add(da, db)), "sub": canon(CTX.subtract(da, db)),
               "mul": canon(CTX.multiply(da, db)), "cmp": int(da.compare(db))}
        try:
            rec["div"] = canon(CTX.divide(da, db))
        except (DivisionByZero, Exception):
            rec["div"] = None
        ops.append(rec)
    # hand-picked HALF_UP edge cases
    for a, b in [("1", "3"), ("2", "3"), ("1", "7"), ("999999999999999", "1"), ("999999999999999", "0.5"),
                 ("999999999999999", "0.4"), ("1", "1e-20"), ("1", "-1e-20"), ("1e30", "999999999999999"),
                 ("1e30", "-999999999999999"), ("100000000000000", "-0.5"), ("100000000000000", "-0.05"),
                 ("100000000000000", "-0.04"), ("123456789012345", "0.5"), ("123456789012345", "0.49"),
                 ("0", "5"), ("5", "0"), ("0", "0"), ("-2.5", "2.5"), ("1e50", "1"), ("1e-9", "1e-9"),
                 ("0.1", "3"), ("4", "0.1"), ("1", "1e-16"), ("1", "-1e-16"), ("1", "5e-15"), ("1", "-5e-16"),
                 ("1", "-4.9e-16"), ("15", "7"), ("1.00000000000001", "-1")]:
        da, db = Decimal(a), Decimal(b)
        rec = {"a": a, "b": b, "add": canon(CTX.add(da, db)), "sub": canon(CTX.subtract(da, db)),
               "mul": canon(CTX.multiply(da, db)), "cmp": int(da.compare(db))}
        rec["div"] = None if db == 0 else canon(CTX.divide(da, db))
        ops.append(rec)
    out["scalar_ops"] = ops
    out["scale6"] = [{"v": s, "text": scale6(Decimal(s))} for s in
                     ["8", "-17", "20", "7.0000004", "7.0000005", "-7.0000005", "0.0000004999", "0.0000005",
                      "123456.7890125", "-0.00000049", "99999.9999995", "1e-20", "563"]]
    out["java_default_name_order"] = {str(n): java_default_name_order(n) for n in (1, 2, 5, 12, 13, 40, 100, 1000)}

    # 2) LP solves ----------------------------------------------------------------------------------
    cases = []
    for (m, n) in [(3, 4), (6, 8), (10, 16), (16, 24), (24, 40)]:
        for r in range(2):
            A, b, c, mx = gen_dense_feasible(rng, m, n)
            cases.append(lp_case("dense_feasible_%dx%d_%d" % (m, n, r), A, b, c, mx, keep_final=(m <= 16)))
    # mixed-sign instances: keep a spread of outcomes
    want = {OPTIMAL: 10, UNBOUNDED: 4, INFEASIBLE: 5, RESTORE_INDEX_FAULT: 3, AUX_UNBOUNDED: 0, NO_DEGENERATE_PIVOT: 0}
    want_p1_opt = 8
    tries = 0
    while tries < 4000 and (sum(want.values()) > 0 or want_p1_opt > 0):
        tries += 1
        m, n = rng.choice([(3, 3), (4, 3), (5, 4), (6, 6), (8, 5), (7, 9), (10, 8)])
        A, b, c, mx = gen_mixed(rng, m, n, neg_b=rng.choice([0.0, 0.3, 0.6]))
        try:
            cs = lp_case("mixed_%dx%d_t%d" % (m, n, tries), A, b, c, mx)
        except Exception:
            continue
        s = cs["status"]
        if s == OPTIMAL and cs["phase1_used"] and want_p1_opt > 0:
            want_p1_opt -= 1
            cases.append(cs)
        elif want.get(s, 0) > 0:
            want[s] -= 1
            cases.append(cs)
    # integer / degenerate instances (ties, zero pivots candidates, degenerate phase-1 endings)
    n_deg = 0
    tries = 0
    have_degenerate_pivot = 0
    while tries < 3000 and (n_deg < 10 or have_degenerate_pivot < 3):
        tries += 1
        m, n = rng.choice([(4, 4), (5, 6), (6, 5), (8, 8)])
        A, b, c, mx = gen_integer_degenerate(rng, m, n)
        try:
            cs = lp_case("intdeg_%dx%d_t%d" % (m, n, tries), A, b, c, mx)
        except Exception:
            continue
        # a degenerate pivot happened iff phase 1 ended with one more phase-1 record after the loop; detect
        # by re-solving is overkill: count instances where phase 1 was used and x0 finished in a slot < n+1
        deg = cs["phase1_used"] and cs["status"] in (OPTIMAL, UNBOUNDED)
        if deg and have_degenerate_pivot < 3:
            have_degenerate_pivot += 1
            cases.append(cs)
        elif n_deg < 10:
            n_deg += 1
            cases.append(cs)
    out["lp_cases"] = cases

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decimal_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    hist = {}
    for cs in cases:
        key = (cs["status"], cs["phase1_used"])
        hist[key] = hist.get(key, 0) + 1
    print("wrote", path, os.path.getsize(path), "bytes;", len(ops), "scalar vectors;", len(cases), "LP cases", hist)


if __name__ == "__main__":
    main()
