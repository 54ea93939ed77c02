"""CPU model of the multi-device decision exchange (lpx_kernels.hip "multi-device decisions", lpx_multi.cpp): one
Python thread per device, numpy row blocks, and exactly the kernel's protocol objects —

  * a mailbox of 2 x n_dev records {ratio, row, a, b_row, tag} per device, slot parity = number of the decision within
    the whole LOOP & 1 (not within the block: a block of odd length would otherwise hand its last slot straight to the
    next block's first decision while a slow peer still reads it — the first version of this test caught that), the
    tag stored LAST, every device storing its candidate into slot `dev` of EVERY device's mailbox and then reducing the
    n_dev records of its own mailbox (lowest global row wins ties);
  * a per-device replica of the pivot-row ring that only the owner of the leaving row fills, and arrival words that
    the other devices wait on before they read it;
  * tags that increase monotonically over the decisions of all blocks.

Random delays between every protocol step shake the interleavings: the two mailbox slots must suffice (a device can
never be more than one decision ahead of a peer), nobody may read a record or a row before its tag / arrival word, and
the final tableau must equal the fp64 oracle's bit for bit.  This is the protocol's logic only — memory-system
visibility on real hardware is what the `-m gpu` tests and the system-scope accesses in the kernel are for."""
import random
import threading
import time

import numpy as np
import pytest

EPS, INF = 1e-9, 1e50


class Device:
    def __init__(self, r, n_dev, A, b, c, row0, m_global, K):
        self.r, self.n_dev, self.row0 = r, n_dev, row0
        self.A, self.b, self.c = A.copy(), b.copy(), c.copy()
        self.m, self.n = A.shape
        self.perm = np.arange(self.n + m_global, dtype=np.int64)
        self.v = 0.0
        self.mail = [[None] * n_dev for _ in range(2)]      # records (ratio, row, a, bi, tag)
        self.prow_ring = np.zeros((K, self.n))
        self.arrive = 0                                     # tag of the last complete row in this replica
        self.status, self.pivots = -1, 0


def run_device(d, devs, K, max_pivots, rng, errors):
    """One device's decision loop: blocks of K decisions from the stale block + pending corrections, then one sweep."""
    try:
        tag = 0
        jitter = lambda: time.sleep(rng.random() * 2e-4) if rng.random() < 0.3 else None
        while d.status == -1:
            pend = []                                       # (e, l_global, p, bl, col_local, prow)
            b_cur = d.b.copy()
            for s in range(K):
                pos = np.nonzero(d.c > EPS)[0]
                if pos.size == 0:
                    d.status = 0
                    break
                e = int(pos[0])
                # phase A: column e of the current tableau = stale column + pending corrections, in order
                a = d.A[:, e].copy()
                for (eu, lu, pu, blu, colu, prowu) in pend:
                    a_new = a - colu * prowu[e]
                    if eu == e:
                        a_new = -(colu / pu)
                    loc = lu - d.row0
                    if 0 <= loc < d.m:
                        a_new[loc] = prowu[e]
                    a = a_new
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = np.where(a < EPS, INF, b_cur / a)
                cand = (INF, 2 ** 31 - 1, 0.0, 0.0)
                if d.m and ratio.min() < INF:
                    i = int(np.argmin(ratio))
                    cand = (float(ratio[i]), d.row0 + i, float(a[i]), float(b_cur[i]))
                tag += 1
                slot = tag & 1                              # alternates over the whole loop
                jitter()
                for peer in devs:                           # store into slot `dev` of every mailbox, tag last
                    peer.mail[slot][d.r] = cand + (tag,)
                    jitter()
                recs = []
                for q in range(d.n_dev):                    # reduce the own mailbox: wait for every record's tag
                    t0 = time.time()
                    while True:
                        rec = d.mail[slot][q]
                        if rec is not None and rec[4] == tag:
                            break
                        assert rec is None or rec[4] < tag, "a peer ran ahead into a live mailbox slot"
                        assert time.time() - t0 < 20, "mailbox wait timed out"
                        time.sleep(0)
                    recs.append(rec)
                win = min(recs, key=lambda r_: (r_[0], r_[1]))
                if not win[0] < INF:
                    d.status = 1
                    break
                if max_pivots >= 0 and d.pivots >= max_pivots:
                    d.status = 9
                    break
                l, p, raw_b = win[1], win[2], win[3]
                owner = d.row0 <= l < d.row0 + d.m
                if owner:                                   # phase B on the owner: row l with the pending pivots applied
                    loc = l - d.row0
                    x = d.A[loc].copy()
                    for (eu, lu, pu, blu, colu, prowu) in pend:
                        if lu == l:
                            x = prowu.copy()
                        else:
                            x_new = x - colu[loc] * prowu
                            x_new[eu] = -(colu[loc] / pu)
                            x = x_new
                    prow = x / p
                    prow[e] = 1.0 / p
                    for peer in devs:                       # broadcast into every replica, then the arrival words
                        peer.prow_ring[s] = prow
                        jitter()
                    for peer in devs:
                        peer.arrive = tag
                else:
                    t0 = time.time()
                    while d.arrive != tag:
                        assert d.arrive < tag, "an owner ran ahead of this device"
                        assert time.time() - t0 < 20, "arrival wait timed out"
                        time.sleep(0)
                prow = d.prow_ring[s].copy()
                pc = d.c[e]
                bl = raw_b / p
                cn = d.c - pc * prow
                cn[e] = -(pc / p)
                d.c = cn
                d.v = d.v + bl * pc
                d.perm[e], d.perm[d.n + l] = d.perm[d.n + l], d.perm[e]
                b_new = b_cur - a * bl
                if owner:
                    b_new[l - d.row0] = bl
                b_cur = b_new
                pend.append((e, l, p, bl, a, prow))
                d.pivots += 1
            # the sweep: every entry through the pending pivots in order, then the special rows / columns
            for (eu, lu, pu, blu, colu, prowu) in pend:
                A_new = d.A - np.outer(colu, prowu)
                A_new[:, eu] = -(colu / pu)
                loc = lu - d.row0
                if 0 <= loc < d.m:
                    A_new[loc] = prowu
                d.A = A_new
            d.b = b_cur
    except Exception as ex:   # pragma: no cover - reported by the main thread
        errors.append((d.r, repr(ex)))
        d.status = 7


@pytest.mark.parametrize("n_dev,shape,K,budget", [(2, (40, 60), 4, -1), (3, (50, 30), 8, 57), (4, (33, 70), 5, -1),
                                                  (8, (64, 40), 3, 40)])
def test_mailbox_protocol_matches_oracle(oracle, n_dev, shape, K, budget):
    m, n = shape
    rng = np.random.default_rng(100 * n_dev + m)
    A, b, c = rng.random((m, n)), (n / 4.0) * (1.0 + rng.random(m)), rng.random(n)
    starts = [(r * m) // n_dev for r in range(n_dev + 1)]                       # LPState.java:222-223
    devs = [Device(r, n_dev, A[starts[r]:starts[r + 1]], b[starts[r]:starts[r + 1]], c, starts[r], m, K)
            for r in range(n_dev)]
    errors = []
    threads = [threading.Thread(target=run_device, args=(d, devs, K, budget, random.Random(7 + d.r), errors)) for d in devs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    ref = oracle.State(A, b, c, kind=oracle.FP64)
    want = ref.simplex_loop(max_pivots=budget)
    wA, wb, wc, wv, wperm = ref.read()
    assert all(d.status == want["status"] and d.pivots == want["pivots"] for d in devs), [(d.status, d.pivots) for d in devs]
    gA = np.vstack([d.A for d in devs])
    gb = np.concatenate([d.b for d in devs])
    u = lambda x: np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    assert np.array_equal(u(gA), u(wA)) and np.array_equal(u(gb), u(wb))
    for d in devs:                                                              # the replicas agree, bit for bit
        assert np.array_equal(u(d.c), u(wc)) and u(np.array([d.v]))[0] == u(np.array([wv]))[0]
        assert list(d.perm) == list(wperm)
