"""LPState: the device-resident slack-form tableau and the three simplex primitives of the reference
(LPState.java:17-320) over the C ABI of liblpx.so.  Same operator names and argument meaning as the
reference so the parity tests read like LPStateSpec.groovy."""
import ctypes as C

import numpy as np

from . import _lib
from .errors import raise_for_status


class LPState:
    def __init__(self, A, b, c, v=0.0, variables=None, coefficients=None, m=None, n=None, device=0,
                 perm=None, row0=0, m_global=None, pricing="reference", block=None, options=None):
        """new LPState(A, b, c, v, variables, coefficients, m, n)  (LPState.java:101-112).
        `variables`/`coefficients` are the reference's name maps (slot -> name / name -> slot); they are
        kept on the host and permuted from the device's slot permutation on demand."""
        L = _lib.lib()
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64)).reshape(-1)
        c = np.ascontiguousarray(np.asarray(c, dtype=np.float64)).reshape(-1)
        self.m = int(b.size if m is None else m)
        self.n = int(c.size if n is None else n)
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if A.size != self.m * self.n:
            if self.m * self.n != 0:    # the C ABI answers LPX_BAD_ARGUMENT to a short array; so does the wrapper
                raise ValueError("LPState: A has %d entries, expected m*n = %d*%d" % (A.size, self.m, self.n))
            A = np.zeros((self.m, self.n))
        A = A.reshape(self.m, self.n)
        self.row0 = int(row0)
        self.m_global = int(self.m if m_global is None else m_global)
        p = None if perm is None else np.ascontiguousarray(np.asarray(perm, dtype=np.int32))
        self._names0 = None             # variable id -> name (the device permutes ids, the host keeps the names)
        if variables is not None and coefficients is not None:
            nslots = self.n + self.m_global
            ids = list(range(nslots)) if p is None else [int(x) for x in p]
            self._names0 = {ids[s]: variables.get(s) for s in range(nslots)}
        h = C.c_void_p()
        rc = L.lpx_state_create(self.m, self.n, A.ctypes.data_as(_lib.dp), max(self.n, 1), b.ctypes.data_as(_lib.dp),
                                c.ctypes.data_as(_lib.dp), float(v), None if p is None else p.ctypes.data_as(_lib.ip),
                                self.row0, self.m_global, int(device), C.byref(h))
        if rc:
            raise_for_status(rc)
        self._h = h
        self._L = L
        if _lib.PRICING[pricing]:   # opt-in Dantzig rule: leaves the reference's pivot sequence on purpose
            rc = L.lpx_state_set_pricing(h, _lib.PRICING[pricing])
            if rc:
                raise_for_status(rc)

        if block is not None:          # pivots per sweep of the device loop: None/0 = by size, 1 = off, 2..64
            rc = L.lpx_state_set_block(h, int(block))
            if rc:
                raise_for_status(rc)
        if _lib.DEFAULT_FUSED is not None and "fused" not in (options or {}):
            self.set_option("fused", int(_lib.DEFAULT_FUSED))
        for key, value in (options or {}).items():
            self.set_option(key, value)

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._L.lpx_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the operator triple
    def get_entering(self):                                      # LPState.java:274
        e = C.c_int32()
        rc = self._L.lpx_get_entering(self._h, C.byref(e))
        if rc:
            raise_for_status(rc)
        return e.value

    def get_leaving(self, entering):                             # LPState.java:287
        l = C.c_int32()
        r = C.c_double()
        rc = self._L.lpx_get_leaving(self._h, int(entering), C.byref(l), C.byref(r))
        if rc:
            raise_for_status(rc)                                 # ValueError <-> IllegalArgumentException
        return l.value

    def pivot(self, entering, leaving):                          # LPState.java:114
        rc = self._L.lpx_pivot(self._h, int(entering), int(leaving))
        if rc:
            raise_for_status(rc)

    getEntering, getLeaving = get_entering, get_leaving

    def simplex_loop(self, max_pivots=-1, track_slot=None):
        """The loop of LPSolver.simplex (LPSolver.java:101-107), device-resident.
        Returns (status, pivots_done, tracked_slot)."""
        piv = C.c_int64()
        st = C.c_int32()
        tr = C.c_int32(-1 if track_slot is None else int(track_slot))
        rc = self._L.lpx_simplex_loop(self._h, int(max_pivots), C.byref(piv), C.byref(st),
                                      C.byref(tr) if track_slot is not None else None)
        if rc:
            raise_for_status(rc)
        return st.value, piv.value, (tr.value if track_slot is not None else None)

    # -- read-back
    def read(self, want_A=True):
        A = np.zeros((self.m, self.n)) if want_A else None
        b = np.zeros(self.m)
        c = np.zeros(self.n)
        v = C.c_double()
        perm = np.zeros(self.n + self.m_global, dtype=np.int32)
        rc = self._L.lpx_state_read(self._h, A.ctypes.data_as(_lib.dp) if want_A and A.size else None, max(self.n, 1),
                                    b.ctypes.data_as(_lib.dp) if self.m else None,
                                    c.ctypes.data_as(_lib.dp) if self.n else None, C.byref(v),
                                    perm.ctypes.data_as(_lib.ip))
        if rc:
            raise_for_status(rc)
        return A, b, c, v.value, perm

    @property
    def A(self):
        return self.read()[0]

    @property
    def b(self):
        return self.read(False)[1]

    @property
    def c(self):
        return self.read(False)[2]

    @property
    def v(self):
        return self.read(False)[3]

    @property
    def perm(self):
        return self.read(False)[4]

    @property
    def variables(self):
        """slot -> name, i.e. the reference's `variables` map after exchangeIndexes (LPState.java:311-320)."""
        if self._names0 is None:
            return None
        perm = self.perm
        return {s: self._names0[int(perm[s])] for s in range(len(perm))}

    @property
    def coefficients(self):
        v = self.variables
        return None if v is None else {name: s for s, name in v.items()}

    def set_option(self, key, value):
        """lpx_state_set_option: tuning / diagnostic option of this handle (names: _lib.OPTIONS)."""
        rc = self._L.lpx_state_set_option(self._h, _lib.OPTIONS[key] if isinstance(key, str) else int(key), int(value))
        if rc:
            raise_for_status(rc)

    def get_option(self, key):
        v = C.c_int64()
        rc = self._L.lpx_state_get_option(self._h, _lib.OPTIONS[key] if isinstance(key, str) else int(key), C.byref(v))
        if rc:
            raise_for_status(rc)
        return v.value

    def info(self):
        """lpx_state_get_info as a dict: what the last loop actually did (block, decision-kernel grid and its
        residency bound, whether the CU-masked streams exist, which XCDs the kernels ran on)."""
        out = _lib.StateInfo()
        rc = self._L.lpx_state_get_info(self._h, C.byref(out))
        if rc:
            raise_for_status(rc)
        d = {k: getattr(out, k) for k, _ in _lib.StateInfo._fields_ if not k.startswith("reserved")}
        d["sweep_kernel_name"] = self._L.lpx_sweep_kernel_name(d["sweep_kernel"]).decode()
        return d

    def chain_trace(self):
        """Phase timestamps (100 MHz ticks) of the last decision launch: array [decisions, 5]; needs option
        chain_trace = 1 before the loop."""
        buf = np.zeros(5 * 64, dtype=np.int64)
        nd = C.c_int32()
        rc = self._L.lpx_state_read_chain_trace(self._h, buf.ctypes.data_as(_lib.i64p), 64, C.byref(nd))
        if rc:
            raise_for_status(rc)
        return buf[: 5 * nd.value].reshape(nd.value, 5)

    def chain_trace_fine(self):
        """Every stamp the decision kernel keeps: array [decisions, 8] for k_block_chain2 (option chain_form = 1: start,
        phase A's loads arrived, candidate published, every candidate read, phase B's loads arrived, hand-off stored,
        phase B done, next entering slot known), [decisions, 5] otherwise."""
        buf = np.zeros(16 * 64, dtype=np.int64)
        nd, ns = C.c_int32(), C.c_int32()
        rc = self._L.lpx_state_read_chain_trace_fine(self._h, buf.ctypes.data_as(_lib.i64p), 64, C.byref(nd), C.byref(ns))
        if rc:
            raise_for_status(rc)
        return buf[: ns.value * nd.value].reshape(nd.value, ns.value)

    def block(self):
        """Pivots per sweep in effect for the device loop (1 = one update pass per pivot)."""
        return int(self._L.lpx_state_get_block(self._h))

    def checksum(self):
        out = (C.c_uint64 * 3)()
        rc = self._L.lpx_state_checksum(self._h, out)
        if rc:
            raise_for_status(rc)
        return int(out[0]), int(out[1]), int(out[2])

    def profile_enable(self, every=1):
        """Bracket every `every`-th row-update launch with HIP events (0/False: off)."""
        rc = self._L.lpx_profile_enable(self._h, int(every))
        if rc:
            raise_for_status(rc)

    def profile_read(self):
        n = C.c_int64()
        ms = C.c_double()
        rc = self._L.lpx_profile_read(self._h, C.byref(n), C.byref(ms))
        if rc:
            raise_for_status(rc)
        return n.value, ms.value


def checksum_host(A, b, c, row0=0):
    """Host restatement of lpx_state_checksum (position-keyed sum of mixed bit patterns, mod 2^64)."""
    def mix(bits, pos):
        with np.errstate(over="ignore"):
            h = bits + (pos + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
            h ^= h >> np.uint64(30)
            h *= np.uint64(0xBF58476D1CE4E5B9)
            h ^= h >> np.uint64(27)
            h *= np.uint64(0x94D049BB133111EB)
            h ^= h >> np.uint64(31)
        return h

    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    with np.errstate(over="ignore"):
        posA = (np.arange(m, dtype=np.uint64)[:, None] + np.uint64(row0)) * np.uint64(n) + np.arange(n, dtype=np.uint64)[None, :]
        sa = int(np.sum(mix(A.view(np.uint64), posA), dtype=np.uint64)) if A.size else 0
        bb = np.ascontiguousarray(b, dtype=np.float64)
        sb = int(np.sum(mix(bb.view(np.uint64), np.arange(bb.size, dtype=np.uint64) + np.uint64(row0)), dtype=np.uint64)) if bb.size else 0
        cc = np.ascontiguousarray(c, dtype=np.float64)
        sc = int(np.sum(mix(cc.view(np.uint64), np.arange(cc.size, dtype=np.uint64)), dtype=np.uint64)) if cc.size else 0
    return sa, sb, sc
