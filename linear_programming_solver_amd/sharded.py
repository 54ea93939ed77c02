"""Row-block sharded simplex loop: one process per GPU, one exchange step per pivot.

Partition: rank r owns the contiguous rows [r*m/G, (r+1)*m/G) of A and b — the partition the reference's
pivotConcurrently uses for its row phase (LPState.java:222-223).  c, v, the slot permutation and the loop
state are replicated and updated identically on every rank.  Per pivot:

    propose   (local)   entering slot from the replicated c; local minimum-ratio candidate; pack
                        {ratio, global row, b[row], raw row} into an (8+n)-double record
    all_gather (RCCL)   the ONE collective of the pivot: G records, (8+n)*8 bytes each (128 KiB at n=16384)
    commit    (local)   every rank picks the same winner (min ratio, lowest row), normalises the pivot row,
                        updates its replicas and runs the row update on its own block

which is allreduce(min+loc) + broadcast(pivot row) folded into a single latency-bound collective with no
data-dependent root, so nothing is ever decided on the host inside the loop.  The host polls the replicated
status word every `poll_every` pivots.
"""
import ctypes as C

import numpy as np

from . import _lib
from .errors import raise_for_status

RUNNING = -1


def row_block(m, nranks, rank):
    """[from, to) of rank's rows: k*m/G .. (k+1)*m/G, as LPState.java:222-223."""
    return (rank * m) // nranks, ((rank + 1) * m) // nranks


class HipShardEngine:
    """One row-block shard on one GPU (liblpx.so lpx_shard_*).  Candidate / gathered records are torch CUDA
    tensors so that torch.distributed (backend "nccl" = RCCL) can move them; kernels are issued on torch's
    current stream."""

    def __init__(self, A_local, b_local, c, row0, m_global, nranks, device=0, perm=None, v=0.0, stream=None,
                 comm_stream=None, reserve_xcds=1, pricing="reference", pipeline=2, fused=None):
        import torch
        self.torch = torch
        L = _lib.lib()
        self._L = L
        A_local = np.ascontiguousarray(A_local, dtype=np.float64)
        b_local = np.ascontiguousarray(b_local, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        self.m_local, self.n = A_local.shape if A_local.ndim == 2 else (0, c.size)
        self.n = c.size
        self.row0, self.m_global, self.nranks = int(row0), int(m_global), int(nranks)
        self.device = int(device)
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        h = C.c_void_p()
        rc = L.lpx_state_create(self.m_local, self.n, A_local.ctypes.data_as(_lib.dp), max(self.n, 1),
                                b_local.ctypes.data_as(_lib.dp), c.ctypes.data_as(_lib.dp), float(v),
                                None if p is None else p.ctypes.data_as(_lib.ip), self.row0, self.m_global,
                                self.device, C.byref(h))
        if rc:
            raise_for_status(rc)
        self._h = h
        if _lib.PRICING[pricing]:
            rc = L.lpx_state_set_pricing(h, _lib.PRICING[pricing])
            if rc:
                raise_for_status(rc)
        want = _lib.DEFAULT_FUSED if fused is None else bool(fused)
        if want is not None:   # every rank alike (None: the library's choice — shards keep the two roundings per update)
            rc = L.lpx_state_set_option(h, _lib.OPTIONS["fused"], int(want))
            if rc:
                raise_for_status(rc)
        dev = torch.device("cuda", self.device)
        rec = _lib.CAND_HEADER + self.n
        self.cand = torch.zeros(rec, dtype=torch.float64, device=dev)
        self.gathered = torch.zeros(rec * self.nranks, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        # Kernels and the collective must be ordered on ONE explicit stream: torch's default stream is the
        # NULL stream, which a non-blocking HIP stream does not synchronise with.
        if stream is not None:
            self.stream = stream
            rc = self._L.lpx_state_set_stream(self._h, C.c_void_p(self.stream.cuda_stream))
        else:
            # the library's own stream with one XCD (32 of 256 CUs) masked out: the HBM-bound row update keeps its
            # rate on 7 XCDs and the collective / decision kernels of the comm stream always find free CUs
            raw = C.c_void_p()
            rc = self._L.lpx_state_use_masked_stream(self._h, int(reserve_xcds), C.byref(raw))
            if not rc:
                self.stream = torch.cuda.ExternalStream(raw.value, device=dev)
        if rc:
            raise_for_status(rc)
        # second stream of the look-ahead pipeline: all-gather + decision of pivot t+1 beside the update of t
        # High priority: a decision kernel of <=16 workgroups must be dispatched between the 10^5 workgroups of
        # the streaming row update instead of queueing behind them (measured: without it the "overlapped"
        # k_commit only completes when k_update drains).
        self.comm_stream = comm_stream if comm_stream is not None else torch.cuda.Stream(device=dev, priority=-1)
        rc = self._L.lpx_shard_set_comm_stream(self._h, C.c_void_p(self.comm_stream.cuda_stream))
        if rc:
            raise_for_status(rc)
        # pipeline 2 = fully overlapped (out-of-place update between two tableau buffers; 2x tableau memory)
        rc = self._L.lpx_shard_set_pipeline(self._h, int(pipeline))
        if rc:
            raise_for_status(rc)

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpx_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self, max_pivots=-1, track_slot=-1):
        rc = self._L.lpx_shard_begin(self._h, int(max_pivots), int(track_slot))
        if rc:
            raise_for_status(rc)

    def propose(self):
        rc = self._L.lpx_shard_propose(self._h, C.c_void_p(self.cand.data_ptr()))
        if rc:
            raise_for_status(rc)

    def commit(self, probe_only=False):
        fn = self._L.lpx_shard_probe if probe_only else self._L.lpx_shard_commit
        rc = fn(self._h, C.c_void_p(self.gathered.data_ptr()), self.nranks)
        if rc:
            raise_for_status(rc)

    # ---- look-ahead form
    def peek(self, slot, pending):
        rc = self._L.lpx_shard_peek(self._h, C.c_void_p(self.cand.data_ptr()), int(slot), 1 if pending else 0)
        if rc:
            raise_for_status(rc)

    def decide(self, slot):
        rc = self._L.lpx_shard_decide(self._h, C.c_void_p(self.gathered.data_ptr()), self.nranks, int(slot))
        if rc:
            raise_for_status(rc)

    def update(self, slot):
        rc = self._L.lpx_shard_update(self._h, int(slot))
        if rc:
            raise_for_status(rc)

    # ---- blocked form: K decisions from the stale tableau, then one K-fold sweep
    def block_peek(self, slot):
        rc = self._L.lpx_shard_block_peek(self._h, C.c_void_p(self.cand.data_ptr()), int(slot))
        if rc:
            raise_for_status(rc)

    def block_decide(self, slot):
        rc = self._L.lpx_shard_block_decide(self._h, C.c_void_p(self.gathered.data_ptr()), self.nranks, int(slot))
        if rc:
            raise_for_status(rc)

    def block_sweep(self, nslots):
        rc = self._L.lpx_shard_block_sweep(self._h, int(nslots))
        if rc:
            raise_for_status(rc)

    def poll(self):
        piv, st = C.c_int64(), C.c_int32()
        rc = self._L.lpx_shard_poll(self._h, C.byref(piv), C.byref(st))
        if rc:
            raise_for_status(rc)
        return piv.value, st.value

    def read(self, want_A=True):
        A = np.zeros((self.m_local, self.n)) if want_A else None
        b = np.zeros(self.m_local)
        c = np.zeros(self.n)
        v = C.c_double()
        perm = np.zeros(self.n + self.m_global, dtype=np.int32)
        rc = self._L.lpx_state_read(self._h, A.ctypes.data_as(_lib.dp) if want_A and A.size else None, max(self.n, 1),
                                    b.ctypes.data_as(_lib.dp) if self.m_local else None,
                                    c.ctypes.data_as(_lib.dp) if self.n else None, C.byref(v),
                                    perm.ctypes.data_as(_lib.ip))
        if rc:
            raise_for_status(rc)
        return A, b, c, v.value, perm

    def checksum(self):
        out = (C.c_uint64 * 3)()
        rc = self._L.lpx_state_checksum(self._h, out)
        if rc:
            raise_for_status(rc)
        return int(out[0]), int(out[1]), int(out[2])

    def profile_enable(self, every=1):
        self._L.lpx_profile_enable(self._h, int(every))

    def profile_read(self):
        n, ms = C.c_int64(), C.c_double()
        self._L.lpx_profile_read(self._h, C.byref(n), C.byref(ms))
        return n.value, ms.value


class DistExchange:
    """The per-pivot collective over torch.distributed: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo"
    in the CPU tests.  One all_gather_into_tensor of (8+n) doubles per rank."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def all_gather(self, engines, comm=False):
        (eng,) = engines
        stream = getattr(eng, "comm_stream" if comm else "stream", None)
        if stream is None:   # CPU engine (gloo)
            self.dist.all_gather_into_tensor(eng.gathered, eng.cand, group=self.group)
        else:
            with eng.torch.cuda.stream(stream):
                self.dist.all_gather_into_tensor(eng.gathered, eng.cand, group=self.group)


class LocalExchange:
    """All shards live in this process (rank-free rehearsal of the protocol on one GPU / in unit tests)."""

    def all_gather(self, engines, comm=False):
        torch = engines[0].torch
        attr = "comm_stream" if comm else "stream"
        stream = getattr(engines[0], attr, None)
        if stream is None:
            cat = torch.cat([e.cand for e in engines])
            for e in engines:
                e.gathered.copy_(cat)
            return
        assert all(getattr(e, attr) is stream for e in engines), "local rehearsal: all shards must share streams"
        with torch.cuda.stream(stream):
            cat = torch.cat([e.cand for e in engines])
            for e in engines:
                e.gathered.copy_(cat)


def sharded_simplex_loop_lookahead(engines, exchange, max_pivots=-1, track_slot=-1, poll_every=16):
    """Same loop, software-pipelined: the candidate of pivot t+1 is computed (peek) from the tableau BEFORE
    the update of pivot t, so the all-gather and the decision of pivot t+1 run on the comm stream while the
    row update of pivot t streams on the main stream.  Results are bit-identical to the plain form.
    Returns (status, pivots, updates_issued)."""
    for e in engines:
        e.begin(max_pivots, track_slot)
    for e in engines:
        e.peek(0, False)
    exchange.all_gather(engines, comm=True)
    for e in engines:
        e.decide(0)
    issued = 0   # row updates issued == decisions issued - 1
    while True:
        burst = poll_every
        if max_pivots >= 0:
            burst = max(0, min(burst, max_pivots - issued))   # decision number max_pivots is the budget probe
        for k in range(burst):
            t = issued + k
            for e in engines:
                e.peek((t + 1) & 1, True)
            for e in engines:
                e.update(t & 1)
            exchange.all_gather(engines, comm=True)
            for e in engines:
                e.decide((t + 1) & 1)
        issued += burst
        polled = [e.poll() for e in engines]
        pivots, status = polled[0]
        assert all(p == polled[0] for p in polled), "replicated loop state diverged: %r" % (polled,)
        if status != RUNNING:
            return status, pivots, issued
        if burst == 0:
            raise RuntimeError("pivot budget exhausted but loop still running")


def sharded_simplex_loop_blocked(engines, exchange, block, max_pivots=-1, track_slot=-1, poll_blocks=4):
    """Blocked pivoting over shards: per block, `block` decisions (each: local candidate from the STALE shard via
    the pending pivots' rank-1 corrections -> all-gather -> identical decision on every rank), then one sweep that
    applies all of them in a single pass over the shard (1/block of the HBM traffic per pivot, bit-identical).
    Returns (status, pivots, decisions_issued)."""
    for e in engines:
        e.begin(max_pivots, track_slot)
    decided = 0
    while True:
        for _ in range(poll_blocks):
            nb = block
            if max_pivots >= 0:
                nb = max(0, min(nb, max_pivots + 1 - decided))   # the last decision of a budgeted run is the probe
            for k in range(nb):
                for e in engines:
                    e.block_peek(k)
                exchange.all_gather(engines)
                for e in engines:
                    e.block_decide(k)
            for e in engines:
                e.block_sweep(nb)
            decided += nb
            if nb == 0:
                break
        polled = [e.poll() for e in engines]
        pivots, status = polled[0]
        assert all(p == polled[0] for p in polled), "replicated loop state diverged: %r" % (polled,)
        if status != RUNNING:
            return status, pivots, decided
        if max_pivots >= 0 and decided >= max_pivots + 1:
            raise RuntimeError("pivot budget exhausted but loop still running")


def sharded_simplex_loop(engines, exchange, max_pivots=-1, track_slot=-1, poll_every=16, lookahead=False, block=1):
    """LPSolver.simplex's loop (LPSolver.java:101-107) over row-block shards.  `engines`: the shard engines
    living in this process (one per rank in production).  Returns (status, pivots, iterations_issued)."""
    if block > 1:
        return sharded_simplex_loop_blocked(engines, exchange, block, max_pivots, track_slot,
                                            poll_blocks=max(1, poll_every // block))
    if lookahead:
        return sharded_simplex_loop_lookahead(engines, exchange, max_pivots, track_slot, poll_every)
    for e in engines:
        e.begin(max_pivots, track_slot)
    issued = 0
    while True:
        burst = poll_every
        if max_pivots >= 0:
            # max_pivots pivots need max_pivots+1 select steps: the last one only reports LIMIT / UNBOUNDED
            burst = max(0, min(burst, max_pivots + 1 - issued))
        for k in range(burst):
            # step number max_pivots (0-based) is the budget probe: it cannot pivot, so it skips the row update
            probe = max_pivots >= 0 and issued + k == max_pivots
            for e in engines:
                e.propose()
            exchange.all_gather(engines)
            for e in engines:
                e.commit(probe_only=probe)
        issued += burst
        polled = [e.poll() for e in engines]
        pivots, status = polled[0]
        assert all(p == polled[0] for p in polled), "replicated loop state diverged: %r" % (polled,)
        if status != RUNNING:
            return status, pivots, issued
        if burst == 0:
            raise RuntimeError("pivot budget exhausted but loop still running")
