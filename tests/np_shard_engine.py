"""TEST INFRASTRUCTURE: a numpy restatement of one row-block shard (the semantics of k_propose / k_commit /
k_update in linear_programming_solver_amd/csrc/lpx_kernels.hip), used to exercise the N>1 protocol of
linear_programming_solver_amd/sharded.py over the `gloo` backend on CPU.  numpy evaluates `a - ce*p` as a
rounded product followed by a rounded difference, i.e. the same unfused arithmetic as the kernels."""
import numpy as np
import torch

EPS, INF = 1e-9, 1e50
RUNNING, OPTIMAL, UNBOUNDED, PIVOT_LIMIT = -1, 0, 1, 9
HEADER = 8


class NumpyShardEngine:
    stream = None   # CPU engine: sharded.DistExchange / LocalExchange take the non-CUDA path

    def __init__(self, A_local, b_local, c, row0, m_global, nranks):
        self.torch = torch
        self.A = np.array(A_local, dtype=np.float64)
        self.b = np.array(b_local, dtype=np.float64)
        self.c = np.array(c, dtype=np.float64)
        self.m_local, self.n = self.A.shape
        self.row0, self.m_global, self.nranks = row0, m_global, nranks
        self.perm = np.arange(self.n + m_global, dtype=np.int32)
        self.v = 0.0
        rec = HEADER + self.n
        self.cand = torch.zeros(rec, dtype=torch.float64)
        self.gathered = torch.zeros(rec * nranks, dtype=torch.float64)
        self.status, self.pivots, self.max_pivots, self.track = RUNNING, 0, -1, -1
        self.e_next = -1

    def _entering(self):
        pos = np.nonzero(self.c > EPS)[0]
        return int(pos[0]) if pos.size else -1

    def begin(self, max_pivots=-1, track_slot=-1):
        self.status, self.pivots, self.max_pivots, self.track = RUNNING, 0, max_pivots, track_slot
        self.e_next = self._entering()
        if self.e_next < 0:
            self.status = OPTIMAL

    def propose(self):
        cand = self.cand.numpy()
        cand[:] = 0.0
        cand[0] = 0.0 if self.status == RUNNING else self.status + 1
        cand[1] = self.e_next
        cand[2], cand[3] = INF, -1.0
        if self.status != RUNNING:
            return
        a = self.A[:, self.e_next]
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where(a < EPS, INF, self.b / a)
        if ratio.size and ratio.min() < INF:
            i = int(np.argmin(ratio))          # first minimum = lowest row among ties
            cand[2], cand[3], cand[4] = ratio[i], self.row0 + i, self.b[i]
            cand[HEADER:] = self.A[i]

    def commit(self, probe_only=False):
        if self.status != RUNNING:
            return
        rec = HEADER + self.n
        g = self.gathered.numpy().reshape(self.nranks, rec)
        best, win = (INF, 2 ** 31 - 1), -1
        for r in range(self.nranks):
            if g[r, 3] >= 0 and (g[r, 2], int(g[r, 3])) < best:
                best, win = (g[r, 2], int(g[r, 3])), r
        if win < 0 or not best[0] < INF:
            self.status = UNBOUNDED
            return
        if 0 <= self.max_pivots <= self.pivots:
            self.status = PIVOT_LIMIT
            return
        assert not probe_only, "the budget probe step must never be able to pivot"
        e, l = self.e_next, best[1]
        raw, raw_b = g[win, HEADER:].copy(), g[win, 4]
        p = raw[e]
        prow = raw / p
        prow[e] = 1.0 / p
        bl = raw_b / p
        ce = self.A[:, e].copy()
        li = l - self.row0
        self._pending = (ce, prow, e, p, bl, li)
        if not getattr(self, "_defer", False):
            self._apply_pending()
        pc = self.c[e]
        self.v = self.v + bl * pc
        cn = self.c - pc * prow
        cn[e] = -(pc / p)
        self.c = cn
        self.perm[e], self.perm[self.n + l] = self.perm[self.n + l], self.perm[e]
        if self.track >= 0:
            if e == self.track:
                self.track = l + self.n
            elif l + self.n == self.track:
                self.track = e
        self.pivots += 1
        self.e_next = self._entering()
        if self.e_next < 0:
            self.status = OPTIMAL

    # ---- look-ahead form.  A sequential engine has nothing to overlap: the decision is stored, the "pending"
    # update is applied when the next peek (or a read) needs the tableau, which is observably identical.
    _pending = None
    comm_stream = None

    def decide(self, slot):
        self._defer = True
        self.commit()
        self._defer = False

    def update(self, slot):
        pass

    def peek(self, slot, pending):
        self._apply_pending()
        self.propose()

    def _apply_pending(self):
        if self._pending is not None:
            ce, prow, e, p, bl, li = self._pending
            self._pending = None
            self.A = self.A - ce[:, None] * prow[None, :]
            self.A[:, e] = -(ce / p)
            self.b = self.b - ce * bl
            if 0 <= li < self.m_local:
                self.A[li] = prow
                self.b[li] = bl

    # ---- blocked form: a sequential engine simply applies the pending pivot before looking again
    def block_peek(self, slot):
        self._apply_pending()
        self.propose()

    def block_decide(self, slot):
        self.decide(slot)

    def block_sweep(self, nslots):
        self._apply_pending()

    def poll(self):
        return self.pivots, self.status

    def read(self, want_A=True):
        self._apply_pending()
        return (self.A if want_A else None), self.b, self.c, self.v, self.perm
