"""LPSolver: `BigDecimal LPSolver.solve(LPStandardForm) throws LPException` (LPSolver.java:78) over
liblpx.so.  Returns the objective as a decimal.Decimal with 6 fractional digits — the analogue of
`v.setScale(6, RoundingMode.HALF_UP)` (LPSolver.java:113) — and raises the reference's exceptions with the
reference's messages."""
import ctypes as C
from decimal import Decimal

import numpy as np

from . import _lib
from .errors import raise_for_status
from .java_compat import hashmap_key_order


class SolveInfo:
    """What the reference only logs: pivot counts, phase-1 use, the unrounded objective, basis, x*."""

    def __init__(self, res, perm, x):
        self.status = res.status
        self.phase1_used = bool(res.phase1_used)
        self.objective = res.objective
        self.objective_text = res.objective_text.decode()
        self.pivots_phase1 = res.pivots_phase1
        self.pivots_phase2 = res.pivots_phase2
        self.x0_slot = res.x0_slot
        self.seconds_total = res.seconds_total
        self.seconds_pivots = res.seconds_pivots
        self.perm = perm
        self.x = x


class LPSolver:
    def __init__(self, device=0, max_pivots=-1, pricing="reference"):
        """pricing="reference": the reference's first-positive rule (default, parity with the Java solver);
        pricing="dantzig": opt-in largest-coefficient rule (same optimum, ~10x fewer pivots, no pivot parity)."""
        self.device = int(device)
        self.max_pivots = int(max_pivots)
        self.pricing = _lib.PRICING[pricing]
        self.last = None

    def solve(self, st_form, restore_order=None):
        """LPSolver.solve(stForm).  Unlike the reference this never modifies `st_form` (the reference
        negates stForm.c in place for `min`, :86-89, and pivots inside stForm.A/b/c, :267)."""
        L = _lib.lib()
        m, n = st_form.m, st_form.n
        A = np.ascontiguousarray(st_form.A, dtype=np.float64)
        b = np.ascontiguousarray(st_form.b, dtype=np.float64)
        c = np.ascontiguousarray(st_form.c, dtype=np.float64)
        opts = _lib.SolveOptions()
        opts.device = self.device
        opts.has_variable_names = 1 if st_form.has_variable_names() else 0
        opts.max_pivots = self.max_pivots
        opts.pricing = self.pricing
        order = None
        if restore_order is not None:
            order = np.ascontiguousarray(np.asarray(restore_order, dtype=np.int32))
        elif st_form.has_variable_names() and n > 0:
            # iteration order of initial.coefficients.keySet() (LPSolver.java:213-217): the keys were put
            # in index order by LPInputReader.processObjective/processConstraint
            names = [st_form.variables[i] for i in range(n)]
            key_order = hashmap_key_order(names)
            order = np.array([st_form.coefficients[k] for k in key_order], dtype=np.int32)
        if order is not None:
            opts.restore_order = order.ctypes.data_as(_lib.ip)
        perm = np.zeros(n + m + 1, dtype=np.int32)
        x = np.zeros(max(n, 1), dtype=np.float64)
        opts.perm_out = perm.ctypes.data_as(_lib.ip)
        opts.x_out = x.ctypes.data_as(_lib.dp)
        res = _lib.SolveResult()
        rc = L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp) if A.size else None, max(n, 1),
                         b.ctypes.data_as(_lib.dp) if m else None, c.ctypes.data_as(_lib.dp) if n else None,
                         1 if st_form.maximize else 0, C.byref(opts), C.byref(res))
        self.last = SolveInfo(res, perm[: n + m].copy(), x[:n].copy())
        if rc != _lib.OPTIMAL:
            raise_for_status(rc)
        return Decimal(res.objective_text.decode())
