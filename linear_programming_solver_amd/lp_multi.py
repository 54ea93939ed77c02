"""LPMulti: the LPState operator triple and the device loop (LPState.java:114-320, LPSolver.java:101-107) with the
row blocks of the tableau on several GPUs of one node, over the C ABI's lpx_multi_* (one process, one handle; the
devices exchange candidates and pivot rows among themselves over xGMI).  Same names and meaning as LPState."""
import ctypes as C

import numpy as np

from . import _lib
from .errors import raise_for_status


class LPMulti:
    def __init__(self, A, b, c, v=0.0, devices=(0,), perm=None, pricing="reference", block=None, options=None):
        L = _lib.lib()
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64)).reshape(-1)
        c = np.ascontiguousarray(np.asarray(c, dtype=np.float64)).reshape(-1)
        self.m, self.n = int(b.size), int(c.size)
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
        if A.size != self.m * self.n:
            raise ValueError("LPMulti: A has %d entries, expected m*n = %d*%d" % (A.size, self.m, self.n))
        A = A.reshape(self.m, self.n)
        self.devices = [int(d) for d in devices]
        dev = np.array(self.devices, dtype=np.int32)
        p = None if perm is None else np.ascontiguousarray(np.asarray(perm, dtype=np.int32))
        h = C.c_void_p()
        rc = L.lpx_multi_create(self.m, self.n, A.ctypes.data_as(_lib.dp), max(self.n, 1), b.ctypes.data_as(_lib.dp),
                                c.ctypes.data_as(_lib.dp), float(v), None if p is None else p.ctypes.data_as(_lib.ip),
                                dev.ctypes.data_as(_lib.ip), len(self.devices), C.byref(h))
        if rc:
            raise_for_status(rc)
        self._h, self._L = h, L
        if _lib.PRICING[pricing]:
            rc = L.lpx_multi_set_pricing(h, _lib.PRICING[pricing])
            if rc:
                raise_for_status(rc)
        if block is not None:
            self.set_option("block", block)
        if _lib.DEFAULT_FUSED is not None and "fused" not in (options or {}):
            self.set_option("fused", int(_lib.DEFAULT_FUSED))
        for key, value in (options or {}).items():
            self.set_option(key, value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpx_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        rc = self._L.lpx_multi_set_option(self._h, _lib.OPTIONS[key] if isinstance(key, str) else int(key), int(value))
        if rc:
            raise_for_status(rc)

    def get_entering(self):                                      # LPState.java:274
        e = C.c_int32()
        rc = self._L.lpx_multi_get_entering(self._h, C.byref(e))
        if rc:
            raise_for_status(rc)
        return e.value

    def get_leaving(self, entering):                             # LPState.java:287
        l, r = C.c_int32(), C.c_double()
        rc = self._L.lpx_multi_get_leaving(self._h, int(entering), C.byref(l), C.byref(r))
        if rc:
            raise_for_status(rc)
        return l.value

    def pivot(self, entering, leaving):                          # LPState.java:114
        rc = self._L.lpx_multi_pivot(self._h, int(entering), int(leaving))
        if rc:
            raise_for_status(rc)

    def simplex_loop(self, max_pivots=-1, track_slot=None):
        """The loop of LPSolver.simplex (LPSolver.java:101-107).  Returns (status, pivots_done, tracked_slot)."""
        piv, st = C.c_int64(), C.c_int32()
        tr = C.c_int32(-1 if track_slot is None else int(track_slot))
        rc = self._L.lpx_multi_simplex_loop(self._h, int(max_pivots), C.byref(piv), C.byref(st),
                                            C.byref(tr) if track_slot is not None else None)
        if rc:
            raise_for_status(rc)
        return st.value, piv.value, (tr.value if track_slot is not None else None)

    def read(self, want_A=True):
        A = np.zeros((self.m, self.n)) if want_A else None
        b, c = np.zeros(self.m), np.zeros(self.n)
        v = C.c_double()
        perm = np.zeros(self.n + self.m, dtype=np.int32)
        rc = self._L.lpx_multi_read(self._h, A.ctypes.data_as(_lib.dp) if want_A and A.size else None, max(self.n, 1),
                                    b.ctypes.data_as(_lib.dp) if self.m else None,
                                    c.ctypes.data_as(_lib.dp) if self.n else None, C.byref(v),
                                    perm.ctypes.data_as(_lib.ip))
        if rc:
            raise_for_status(rc)
        return A, b, c, v.value, perm

    @property
    def v(self):
        return self.read(False)[3]

    def checksum(self):
        out = (C.c_uint64 * 3)()
        rc = self._L.lpx_multi_checksum(self._h, out)
        if rc:
            raise_for_status(rc)
        return int(out[0]), int(out[1]), int(out[2])

    def info(self):
        out = _lib.StateInfo()
        rc = self._L.lpx_multi_get_info(self._h, C.byref(out))
        if rc:
            raise_for_status(rc)
        d = {k: getattr(out, k) for k, _ in _lib.StateInfo._fields_ if not k.startswith("reserved")}
        d["sweep_kernel_name"] = self._L.lpx_sweep_kernel_name(d["sweep_kernel"]).decode()
        return d

    def block(self):
        return self.info()["block"]

    def profile_enable(self, every=1):
        rc = self._L.lpx_multi_profile_enable(self._h, int(every))
        if rc:
            raise_for_status(rc)

    def profile_read(self, shard=0):
        n, ms = C.c_int64(), C.c_double()
        rc = self._L.lpx_multi_profile_read(self._h, int(shard), C.byref(n), C.byref(ms))
        if rc:
            raise_for_status(rc)
        return n.value, ms.value
