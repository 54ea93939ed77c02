"""CPU model of the decision kernel's pending-pivot ladder (k_block_chain2_t): the entering column and the
leaving row of the CURRENT tableau are recovered from a STALE tableau plus the pending pivots' own columns and rows, each
entry's chain starting behind the LAST pending pivot that replaced it (the pivot that entered at the column's slot / left
through the entry's row for a column; the pivot that left through the row / entered at the entry's slot for a row).  The
model applies exactly the operations of LPState.java:139-164 in the plain arithmetic (product and difference rounded
separately) and must reproduce the fp64 oracle's tableau bit for bit — the claim the GPU ladder rests on, checked without
a GPU."""
import numpy as np
import pytest


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _run(oracle, m, n, seed, pivots):
    rng = np.random.default_rng(seed)
    A, b, c = rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)
    st = oracle.State(A, b, c, kind=oracle.FP64)
    tabs = [st.read()[0].copy()]
    piv = []
    for _ in range(pivots):
        e = st.get_entering()
        if e < 0:
            break
        l = st.get_leaving(e)
        if l < 0:
            break
        before = tabs[-1]
        p = before[l, e]
        st.pivot(e, l)
        after = st.read()[0].copy()
        # what the decision kernel keeps of a pivot: the entering column BEFORE the pivot, the normalised pivot row AFTER it
        piv.append({"e": e, "l": l, "p": p, "col": before[:, e].copy(), "prow": after[l, :].copy()})
        tabs.append(after)
    st.close()
    return tabs, piv


def _column_by_ladder(stale, pend, e):
    """column e after the pending pivots, from the stale tableau: per row the chain starts behind max(ra, rs)"""
    m = stale.shape[0]
    ra = max([u for u, q in enumerate(pend) if q["e"] == e], default=-1)
    out = np.empty(m)
    for i in range(m):
        rs = max([u for u, q in enumerate(pend) if q["l"] == i], default=-1)
        if rs >= ra and rs >= 0:
            x = pend[rs]["prow"][e]                       # the row became the normalised pivot row (:139-145)
        elif ra >= 0:
            x = -(pend[ra]["col"][i] / pend[ra]["p"])     # the column was replaced by -(col / p) (:157)
        else:
            x = stale[i, e]
        for u in range(max(ra, rs) + 1, len(pend)):
            x = x - pend[u]["col"][i] * pend[u]["prow"][e]   # two roundings, LPState.java:162
        out[i] = x
    return out


def _row_by_ladder(stale, pend, l):
    n = stale.shape[1]
    rb = max([u for u, q in enumerate(pend) if q["l"] == l], default=-1)
    out = np.empty(n)
    for j in range(n):
        rs = max([u for u, q in enumerate(pend) if q["e"] == j], default=-1)
        if rs > rb:
            x = -(pend[rs]["col"][l] / pend[rs]["p"])     # row l's entry of the column that entered at slot j (:157)
        elif rb >= 0:
            x = pend[rb]["prow"][j]                       # the row restarts from the pivot row it became
        else:
            x = stale[l, j]
        for u in range(max(rb, rs) + 1, len(pend)):
            x = x - pend[u]["col"][l] * pend[u]["prow"][j]
        out[j] = x
    return out


@pytest.mark.parametrize("m,n,seed", [(24, 40, 1), (40, 24, 2), (16, 64, 3)])
def test_ladder_with_start_indices_reproduces_the_oracle(oracle, m, n, seed):
    tabs, piv = _run(oracle, m, n, seed, 90)
    assert len(piv) >= 40
    restarts = hits = 0
    for k in range(8, len(piv), 3):
        for pending in sorted({min(k, 5), min(k, 17), min(k, 48)}):
            stale, pend = tabs[k - pending], piv[k - pending:k]
            cur = tabs[k]
            e, l = piv[k]["e"], piv[k]["l"]
            restarts += any(q["e"] == e for q in pend)
            hits += any(q["l"] == l for q in pend)
            assert np.array_equal(_bits(_column_by_ladder(stale, pend, e)), _bits(cur[:, e])), (k, pending, "column")
            assert np.array_equal(_bits(_row_by_ladder(stale, pend, l)), _bits(cur[l, :])), (k, pending, "row")
    # the case the ladder was rebuilt for is the common one under the first-positive rule
    assert restarts >= 10 and hits >= 10, (restarts, hits)
